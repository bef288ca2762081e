#!/bin/bash
# tools/build_variant.sh <tag> "<-D flags>" -- build the WORKING TREE's library with extra -D switches into dasp_amd/variants/<tag>/libdasp_amd.so
# (git-ignored, travels to the GPU box; probes load it through DASP_AMD_SO).  Experiment switches never reach the product build.
set -e
tag=$1; flags=$2
root=$(cd "$(dirname "$0")/.." && pwd)
mkdir -p "$root/dasp_amd/variants/$tag"
make -C "$root/dasp_amd/csrc" -s -j8 OBJDIR="build_$tag" OUT="../variants/$tag/libdasp_amd.so" EXTRA="$flags" "../variants/$tag/libdasp_amd.so"
echo "built dasp_amd/variants/$tag/libdasp_amd.so with $flags"

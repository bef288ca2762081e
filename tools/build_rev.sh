#!/bin/bash
# tools/build_rev.sh <git-rev> <tag> -- build libdasp_amd.so + dasp_bench of another revision into dasp_amd/variants/<tag>/
# (git-ignored, travels to the GPU box) so that tools/ab.sh can compare two revisions on one device.  Rejected kernel variants
# live in git history / DESIGN.md 4.4, not behind macros in kernels.hip.
set -e
rev=$1; tag=$2
root=$(cd "$(dirname "$0")/.." && pwd)
wt=$(mktemp -d /tmp/dasp_wt.XXXXXX)
git -C "$root" worktree add --detach "$wt" "$rev" > /dev/null
make -C "$wt/dasp_amd/csrc" -s -j8 ../libdasp_amd.so ../bin/dasp_bench
mkdir -p "$root/dasp_amd/variants/$tag"
cp "$wt/dasp_amd/libdasp_amd.so" "$root/dasp_amd/variants/$tag/"
# re-link the driver against the copy next to it
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC -pthread "$wt/dasp_amd/csrc/cli_bench.cpp" -o "$root/dasp_amd/variants/$tag/dasp_bench" \
  -L"$root/dasp_amd/variants/$tag" -ldasp_amd -Wl,-rpath,'$ORIGIN'
git -C "$root" worktree remove --force "$wt"
echo "built dasp_amd/variants/$tag from $rev"

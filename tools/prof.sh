#!/bin/bash
# tools/prof.sh -- rocprofv3 passes on the GPU box (run through gpurun from the repo root).
# usage: tools/prof.sh <tag> -- <program> [args...]     (the program itself after --, never a wrapper)
# Produces under gpurun_out/prof_<tag>/: kernel-trace + stats CSVs, and two PMC passes
# (FETCH_SIZE, WRITE_SIZE in separate runs: they do not fit one pass on gfx950).
set -u
tag=$1; shift; shift
root=${GRAFT_REPO_ROOT:-$PWD}
out=$root/gpurun_out/prof_$tag
mkdir -p "$out"
export TMPDIR=/tmp
cd "$root"
timeout -k 5 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$out/trace" -- "$@" > "$out/trace.log" 2>&1
timeout -k 5 600 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d "$out/pmc_fetch" -- "$@" > "$out/pmc_fetch.log" 2>&1
timeout -k 5 600 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d "$out/pmc_write" -- "$@" > "$out/pmc_write.log" 2>&1
find "$out" -name "*.csv" | head -20

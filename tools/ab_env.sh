#!/bin/bash
# tools/ab_env.sh "<workload args>" "<ENV=val ...>" ... -- interleaved A/B of ONE dasp_bench under different environments (2 rounds)
args=$1; shift
for round in 1 2; do
  for e in "$@"; do
    printf "%s | %s | " "$e" "$args"
    env $e timeout 300 dasp_amd/bin/dasp_bench $args 2>&1 | tail -1 | sed -E 's/.*\| ([0-9.]+ ms \(event [0-9.]+\)).*alg = ([0-9.]+) of.*graph: ([0-9.]+) ms.*mismatches=([0-9]+)/\1 frac=\2 graph_ms=\3 bad=\4/'
  done
done

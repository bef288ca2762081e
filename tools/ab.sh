#!/bin/bash
# tools/ab.sh -- interleaved A/B of dasp_bench builds on one device (rule 24: same process order, same box)
# usage: tools/ab.sh "<workload args>" <bench1> <bench2> ...   ; 3 rounds each, prints the event ms per round
# (other revisions: tools/build_rev.sh <rev> <tag> -> dasp_amd/variants/<tag>/dasp_bench)
args=$1; shift
for round in 1 2 3; do
  for b in "$@"; do
    printf "%s | %s | " "$b" "$args"
    timeout 300 $b $args 2>&1 | tail -1 | sed -E 's/.*\| ([0-9.]+ ms \(event [0-9.]+\)).*alg = ([0-9.]+) of.*graph: ([0-9.]+) ms.*mismatches=([0-9]+)/\1 frac=\2 graph_ms=\3 bad=\4/'
  done
done

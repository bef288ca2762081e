#!/bin/bash
# tools/sweep.sh -- the sweeps behind the automatic choices (profiles/r01_auto_rule_sweeps.md); run through gpurun from the repo root
B=dasp_amd/bin/dasp_bench
short() { sed -E "s/ rows=([0-9]+) nnz=([0-9]+).*win=([0-9\/]+) lds=([0-9]+)B c16=([0-9]) panels=([0-9]+) \| ([0-9.]+) ms.*mismatches=([0-9]+)/ nnz=\2 win=\3 c16=\5 panels=\6 ms=\7 bad=\8/"; }
echo "== scale sweep, defaults"
for w in cop20k_A nlpkkt160 powerlaw_1M Queen_4147 webbase-1M ljournal-2008 HV15R; do
  for sc in 0.03 0.11 0.27 0.41 0.63 0.88; do timeout 300 $B $w $sc 64 20 3 | tail -1 | short; done; done
echo "== cid16 x cache policy"
for w in HV15R nlpkkt160 Queen_4147; do for sc in 0.03 0.05 0.08 0.12; do for cfg in "-1 1" "1 1" "-1 2" "1 2" "0 0"; do set -- $cfg
  timeout 200 $B $w $sc 64 300 30 0.75 0 0 0 $1 0 $2 | tail -1 | short | sed "s/^/cid16=$1 policy=$2 /"; done; done; done
echo "== window height"
for sc in 0.1 0.3 0.6 1 1.5 2 4 8; do for rw in 0 128 256 512 768 1024; do
  timeout 200 $B cop20k_A $sc 64 500 50 0.75 0 0 $rw | tail -1 | short | sed "s/^/rw=$rw /"; done; done
echo "== column panels"
for w in "ljournal-2008 0.27 64" "ljournal-2008 0.41 16" "powerlaw_1M 0.63 64" "powerlaw_1M 1 64" "ljournal-2008 1 16"; do for cp in 1 0 2 3 4; do
  timeout 300 $B $w 100 10 0.75 0 0 0 0 $cp | tail -1 | short | sed "s/^/col_panels=$cp /"; done; done

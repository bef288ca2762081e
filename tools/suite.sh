#!/bin/bash
# tools/suite.sh -- the seven stand-ins through the torch-free driver (run via gpurun from the repo root)
B=dasp_amd/bin/dasp_bench
for w in "cop20k_A 1 64 1000 100" "HV15R 1 64 100 10" "Queen_4147 1 64 100 10" "nlpkkt160 1 64 100 10" "powerlaw_1M 1 64 200 20" "webbase-1M 1 16 1000 100" "ljournal-2008 1 16 200 20"; do
  timeout 300 $B $w 2>&1 | tail -1
done

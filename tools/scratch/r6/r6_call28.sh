#!/bin/bash
# r6_call28 -- the automatic-rule guard twice more on one box (is a 7 % gate stable from run to run?)
export PYTHONPATH=$PWD
for r in 1 2; do timeout 900 python3 -m pytest tests/test_zz_auto_rules.py -m gpu -x -q 2>&1 | tail -3; grep -c LOSS gpurun_out/r6_auto_rules.md; cp gpurun_out/r6_auto_rules.md gpurun_out/r6/auto_rules28_$r.md; done

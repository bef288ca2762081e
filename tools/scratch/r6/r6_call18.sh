#!/bin/bash
# r6_call18 -- the f16 dominant-length slab rule against r5's probe of rows of 12..24 (tools/scratch/f16_slab24_probe.py) + the guard
out=gpurun_out/r6; mkdir -p $out
export PYTHONPATH=$PWD
timeout 1200 python3 tools/scratch/f16_slab24_probe.py 2>&1 | grep -v amdgpu.ids > $out/slab24_18.log; cat $out/slab24_18.log
timeout 1500 python3 -m pytest tests/test_zz_auto_rules.py -m gpu -x -q > $out/autorules18.log 2>&1; tail -4 $out/autorules18.log
cp gpurun_out/r6_auto_rules.md $out/auto_rules18.md; grep LOSS $out/auto_rules18.md

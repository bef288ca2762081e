#!/bin/bash
# r6_call14 -- both r5 sweeps again at the r6 sources (profiles/r06_size_sweep.md, r06_category_sweep.md)
out=gpurun_out/r6; mkdir -p $out
export PYTHONPATH=$PWD
timeout 1500 python3 tools/size_sweep.py $out/size_sweep14.md 2>&1 | grep -v amdgpu.ids > $out/size_sweep14.log
timeout 1500 python3 tools/category_sweep.py $out/category_sweep14.md 2>&1 | grep -v amdgpu.ids > $out/category_sweep14.log
tail -50 $out/size_sweep14.log; tail -30 $out/category_sweep14.log

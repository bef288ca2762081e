#!/bin/bash
mkdir -p gpurun_out/r4
python -m pytest tests -m gpu -x -q > gpurun_out/r4/b15_tests.log 2>&1; grep -E "passed|failed" gpurun_out/r4/b15_tests.log | tail -2
timeout 900 python tools/short_seg_ab.py > gpurun_out/r4/b15_short_seg.log 2>&1; grep -v amdgpu gpurun_out/r4/b15_short_seg.log

#!/bin/bash
mkdir -p gpurun_out/r4
python -m pytest tests -m gpu -x -q > gpurun_out/r4/b14_tests.log 2>&1; grep -E "passed|failed" gpurun_out/r4/b14_tests.log | tail -2
for w in HV15R Queen_4147; do
for pc in 2 3 4; do
echo "== $w two-plan fused step, waiting workgroups at 0.8, $pc per CU"
DASP_MG_POLL_PER_CU=$pc PROBE_FULL=0 PROBE_EXCHANGE=push PROBE_AG_US=0,30 timeout 900 python tools/mg_step_probe.py 8 $w 3 2>&1 | grep -v amdgpu.ids | tail -1 | cut -c1-200
done; done

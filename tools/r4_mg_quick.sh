#!/bin/bash
mkdir -p gpurun_out/r4
python -m pytest tests -m gpu -x -q -k "one_stream or fused_mg or mg_spmv" 2>&1 | grep -E "passed|failed"
for w in HV15R Queen_4147; do
PROBE_FULL=1 PROBE_EXCHANGE=push PROBE_AG_US=0,30 timeout 1700 python tools/mg_step_probe.py 8 $w 0,3,7 > gpurun_out/r4/mg_quick_v1_$w.log 2>&1; grep -v amdgpu gpurun_out/r4/mg_quick_v1_$w.log | tail -8 | cut -c1-330
PROBE_FULL=0 PROBE_OVERLAP=2 PROBE_EXCHANGE=push PROBE_AG_US=0,30 timeout 1700 python tools/mg_step_probe.py 8 $w 0,3,7 > gpurun_out/r4/mg_quick_v2_$w.log 2>&1; grep -v amdgpu gpurun_out/r4/mg_quick_v2_$w.log | tail -3 | cut -c1-200
done

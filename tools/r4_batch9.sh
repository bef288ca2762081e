#!/bin/bash
mkdir -p gpurun_out/r4
timeout 600 python -m pytest tests -m gpu -x -q -k "one_stream or fused_mg" > gpurun_out/r4/b9_tests.log 2>&1; tail -3 gpurun_out/r4/b9_tests.log
export DASP_AMD_SO=$PWD/dasp_amd/variants/exp/libdasp_amd.so
for w in HV15R Queen_4147; do
for f in 0.5 0.7 0.85 1.0; do
echo "== $w hot_at $f"
DASP_MG_HOT_AT=$f PROBE_FULL=0 PROBE_OVERLAP=2 PROBE_EXCHANGE=push PROBE_AG_US=0,30,45 timeout 900 python tools/mg_step_probe.py 8 $w 3 2>&1 | grep -v amdgpu.ids | tail -1 | cut -c1-330
done
done

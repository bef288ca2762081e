#!/bin/bash
mkdir -p gpurun_out/r4
timeout 600 python -m pytest tests -m gpu -x -q -k "one_stream or fused_mg or mg_spmv" 2>&1 | tail -2
for w in HV15R Queen_4147; do
for at in 1.0 0.8 0.6; do for pc in 1 2; do
echo "== $w two-plan fused step, waiting workgroups at $at of the grid, $pc per CU"
DASP_MG_POLL_AT=$at DASP_MG_POLL_PER_CU=$pc PROBE_FULL=0 PROBE_EXCHANGE=push PROBE_AG_US=0,15,30 timeout 900 python tools/mg_step_probe.py 8 $w 3 2>&1 | grep -v amdgpu.ids | tail -1 | cut -c1-260
done; done; done

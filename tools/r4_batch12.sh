#!/bin/bash
mkdir -p gpurun_out/r4
export DASP_AMD_SO=$PWD/dasp_amd/variants/exp/libdasp_amd.so
for w in HV15R Queen_4147; do
echo "== $w one-stream step, marked workgroups in place"
PROBE_FULL=0 PROBE_OVERLAP=2 PROBE_EXCHANGE=push PROBE_AG_US=0,15,30 timeout 900 python tools/mg_step_probe.py 8 $w 3 2>&1 | grep -v amdgpu.ids | tail -1 | tee gpurun_out/r4/b12_${w}_inplace.log
echo "== $w one-stream step, bounded persistent workgroups at the end"
DASP_MG_STEP2_POLLERS=1 PROBE_FULL=0 PROBE_OVERLAP=2 PROBE_EXCHANGE=push PROBE_AG_US=0,15,30 timeout 900 python tools/mg_step_probe.py 8 $w 3 2>&1 | grep -v amdgpu.ids | tail -1 | tee gpurun_out/r4/b12_${w}_pollers.log
echo "== $w two-plan fused step (r3)"
PROBE_FULL=0 PROBE_EXCHANGE=push PROBE_AG_US=0,15,30 timeout 900 python tools/mg_step_probe.py 8 $w 3 2>&1 | grep -v amdgpu.ids | tail -1 | tee gpurun_out/r4/b12_${w}_v1.log
done

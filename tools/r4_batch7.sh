#!/bin/bash
mkdir -p gpurun_out/r4
export DASP_AMD_SO=$PWD/dasp_amd/variants/exp/libdasp_amd.so
for w in HV15R Queen_4147; do
PROBE_FULL=0 PROBE_OVERLAP=2 PROBE_EXCHANGE=push PROBE_AG_US=0 timeout 900 python tools/mg_step_probe.py 8 $w 3 > gpurun_out/r4/b7_mg2_$w.log 2>&1; grep -v amdgpu.ids gpurun_out/r4/b7_mg2_$w.log | tail -2
done

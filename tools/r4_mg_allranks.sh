#!/bin/bash
# every rank of the 8-way partitions of HV15R and Queen_4147, one after the other on this one GPU, with the direct exchange in loopback (scratch memory of this
# GPU as the peers, the flags raised N us after the stores for the links' share): the two-plan fused step (r3 + r4 changes) and the one-stream step (r4)
mkdir -p gpurun_out/r4
for w in HV15R Queen_4147; do
PROBE_FULL=1 PROBE_EXCHANGE=push PROBE_AG_US=0,15,30,45 timeout 1700 python tools/mg_step_probe.py 8 $w all > gpurun_out/r4/mg_allranks_v1_$w.log 2>&1; grep -v amdgpu gpurun_out/r4/mg_allranks_v1_$w.log | tail -12 | cut -c1-250
PROBE_FULL=1 PROBE_OVERLAP=2 PROBE_EXCHANGE=push PROBE_AG_US=0,15,30,45 timeout 1700 python tools/mg_step_probe.py 8 $w all > gpurun_out/r4/mg_allranks_v2_$w.log 2>&1; grep -v amdgpu gpurun_out/r4/mg_allranks_v2_$w.log | tail -7 | cut -c1-250
done

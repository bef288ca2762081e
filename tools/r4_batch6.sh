#!/bin/bash
mkdir -p gpurun_out/r4
timeout 900 python -m pytest tests -m gpu -x -q -k "one_stream or fused_mg or mg_spmv or bench" > gpurun_out/r4/b6_tests.log 2>&1; tail -15 gpurun_out/r4/b6_tests.log
for w in HV15R Queen_4147; do
PROBE_FULL=0 PROBE_OVERLAP=2 PROBE_EXCHANGE=push PROBE_AG_US=0,40 timeout 900 python tools/mg_step_probe.py 8 $w 0,3 > gpurun_out/r4/b6_mg2_$w.log 2>&1; grep -v amdgpu.ids gpurun_out/r4/b6_mg2_$w.log | tail -4
PROBE_FULL=0 PROBE_EXCHANGE=push PROBE_AG_US=0,40 timeout 900 python tools/mg_step_probe.py 8 $w 3 > gpurun_out/r4/b6_mg1_$w.log 2>&1; grep -v amdgpu.ids gpurun_out/r4/b6_mg1_$w.log | tail -2
done

#!/bin/bash
# GPU batch 2 (r4): y-vector trials feasibility, fine-grained x cost, step-kernel breakdown
mkdir -p gpurun_out/r4
export DASP_AMD_SO=$PWD/dasp_amd/variants/exp/libdasp_amd.so
timeout 600 python tools/y_trials_probe.py HV15R 8 2 > gpurun_out/r4/b2_ytrials_hv15r.log 2>&1; cat gpurun_out/r4/b2_ytrials_hv15r.log
timeout 600 python tools/y_trials_probe.py nlpkkt160 8 2 > gpurun_out/r4/b2_ytrials_nlpkkt.log 2>&1; cat gpurun_out/r4/b2_ytrials_nlpkkt.log
PROBE_FULL=0 PROBE_AG_US=0 timeout 900 python tools/mg_step_probe.py 8 Queen_4147 3 > gpurun_out/r4/b2_mg_queen_r3.log 2>&1; cat gpurun_out/r4/b2_mg_queen_r3.log
PROBE_FULL=0 PROBE_AG_US=0 timeout 900 python tools/mg_step_probe.py 8 HV15R 3 > gpurun_out/r4/b2_mg_hv15r_r3.log 2>&1; cat gpurun_out/r4/b2_mg_hv15r_r3.log

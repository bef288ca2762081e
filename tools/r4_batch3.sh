#!/bin/bash
mkdir -p gpurun_out/r4
export DASP_AMD_SO=$PWD/dasp_amd/variants/exp/libdasp_amd.so
timeout 600 python tools/placement_cure_probe.py HV15R 4 > gpurun_out/r4/b3_cure_hv15r.log 2>&1; grep -v amdgpu.ids gpurun_out/r4/b3_cure_hv15r.log | grep "pair\|plan"
timeout 600 python tools/placement_cure_probe.py nlpkkt160 3 > gpurun_out/r4/b3_cure_nlpkkt.log 2>&1; grep "pair " gpurun_out/r4/b3_cure_nlpkkt.log
timeout 600 python tools/y_spacer_probe.py HV15R > gpurun_out/r4/b3_spacer_hv15r.log 2>&1; grep candidate gpurun_out/r4/b3_spacer_hv15r.log
timeout 600 python tools/y_spacer_probe.py nlpkkt160 > gpurun_out/r4/b3_spacer_nlpkkt.log 2>&1; grep candidate gpurun_out/r4/b3_spacer_nlpkkt.log

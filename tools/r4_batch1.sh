#!/bin/bash
# GPU batch 1 (r4): baseline tests, placement-cure probes, Queen_4147 8-way step probe
mkdir -p gpurun_out/r4
python -m pytest tests -m gpu -x -q > gpurun_out/r4/b1_tests.log 2>&1; tail -3 gpurun_out/r4/b1_tests.log
DASP_AMD_SO=$PWD/dasp_amd/variants/ystore/libdasp_amd.so timeout 600 python tools/placement_cure_probe.py HV15R 5 > gpurun_out/r4/b1_cure_hv15r.log 2>&1; cat gpurun_out/r4/b1_cure_hv15r.log
DASP_AMD_SO=$PWD/dasp_amd/variants/ystore/libdasp_amd.so timeout 600 python tools/placement_cure_probe.py nlpkkt160 4 > gpurun_out/r4/b1_cure_nlpkkt.log 2>&1; cat gpurun_out/r4/b1_cure_nlpkkt.log
PROBE_FULL=1 PROBE_AG_US=0,40 timeout 1800 python tools/mg_step_probe.py 8 Queen_4147 all > gpurun_out/r4/b1_mg_queen8.log 2>&1; cat gpurun_out/r4/b1_mg_queen8.log

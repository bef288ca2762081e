#!/bin/bash
mkdir -p gpurun_out/r4
python -m pytest tests -m gpu -x -q > gpurun_out/r4/b5_tests.log 2>&1; tail -3 gpurun_out/r4/b5_tests.log
echo "== product build"; timeout 600 python tools/placement_cure_probe.py HV15R 3 2>&1 | grep "plan [0-9]:"
export DASP_AMD_SO=$PWD/dasp_amd/variants/exp/libdasp_amd.so
echo "== experiment build"
for w in HV15R nlpkkt160; do
timeout 600 python tools/placement_cure_probe.py $w 3 > gpurun_out/r4/b5_cure_$w.log 2>&1; echo "== $w"; grep "pair\|plan [0-9]:" gpurun_out/r4/b5_cure_$w.log | grep -v offsets
done

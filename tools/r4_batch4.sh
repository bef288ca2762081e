#!/bin/bash
mkdir -p gpurun_out/r4
export DASP_AMD_SO=$PWD/dasp_amd/variants/exp/libdasp_amd.so
for w in HV15R nlpkkt160 Queen_4147; do
PROBE_XCD_LAST=1 timeout 600 python tools/placement_cure_probe.py $w 3 > gpurun_out/r4/b4_cure_$w.log 2>&1; echo "== $w"; grep "pair\|plan [0-9]:" gpurun_out/r4/b4_cure_$w.log | grep -v offsets
done

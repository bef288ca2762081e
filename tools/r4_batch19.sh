#!/bin/bash
export DASP_AMD_SO=$PWD/dasp_amd/variants/exp/libdasp_amd.so
for w in HV15R nlpkkt160 Queen_4147; do echo "== $w"; timeout 600 python tools/placement_cure_probe.py $w 2 2>&1 | grep "pair [01]:"; done

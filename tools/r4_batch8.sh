#!/bin/bash
mkdir -p gpurun_out/r4
export DASP_AMD_SO=$PWD/dasp_amd/variants/exp/libdasp_amd.so
export TMPDIR=/tmp
rm -rf gpurun_out/r4/b8_prof
R=$GRAFT_REPO_ROOT
(cd /tmp && PROBE_FULL=0 PROBE_OVERLAP=2 PROBE_EXCHANGE=push PROBE_AG_US=0 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/r4/b8_prof -- python3 $R/tools/mg_step_probe.py 8 HV15R 3 > $R/gpurun_out/r4/b8_prof.log 2>&1)
grep -v amdgpu.ids gpurun_out/r4/b8_prof.log | tail -2 | cut -c1-400
f=$(find gpurun_out/r4/b8_prof -name "*kernel_stats.csv" | head -1); head -12 $f | cut -c1-300

#!/bin/bash
# tools/round_end.sh -- the measurement batch behind profiles/r04_*: run through gpurun from the repo root, then `python tools/round_end_collect.py` here.
#   full -m gpu suite | PMC traffic of every workload (tools/traffic_all.sh) | bench.py (full record) | rocprofv3 --kernel-trace --stats of the bench command |
#   every rank of the 8-way HV15R and Queen_4147 partitions with the direct exchange in loopback: two-plan fused step and one-stream step
export ROUND=${ROUND:-r04}
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
python -m pytest tests -m gpu -x -q > gpurun_out/round_end_tests.log 2>&1; grep -E "passed|failed" gpurun_out/round_end_tests.log | tail -1
bash tools/traffic_all.sh > gpurun_out/round_end_traffic.log 2>&1
# profiles/traffic.json at this build's kernel revision BEFORE the bench runs (the bench attaches an entry only when the revisions match); copies travel back in gpurun_out/
python tools/traffic_collect.py > gpurun_out/round_end_traffic_collect.log 2>&1 && cp profiles/traffic.json gpurun_out/round_end_traffic.json && cp profiles/${ROUND}_traffic.md gpurun_out/round_end_traffic.md
python bench.py > gpurun_out/round_end_bench.json.log 2> gpurun_out/round_end_bench.err
export TMPDIR=/tmp
rm -rf gpurun_out/round_end_prof
(cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/round_end_prof -- python3 $GRAFT_REPO_ROOT/bench.py --no-suite --no-vendor --steps 20 > $GRAFT_REPO_ROOT/gpurun_out/round_end_prof.log 2>&1)
for w in HV15R Queen_4147; do
PROBE_FULL=1 PROBE_EXCHANGE=push PROBE_AG_US=0,15,30,45 timeout 1700 python tools/mg_step_probe.py 8 $w all > gpurun_out/round_end_mg_v1_$w.log 2>&1; tail -5 gpurun_out/round_end_mg_v1_$w.log | cut -c1-200
PROBE_FULL=1 PROBE_OVERLAP=2 PROBE_EXCHANGE=push PROBE_AG_US=0,15,30,45 timeout 1700 python tools/mg_step_probe.py 8 $w all > gpurun_out/round_end_mg_v2_$w.log 2>&1; tail -5 gpurun_out/round_end_mg_v2_$w.log | cut -c1-200
done

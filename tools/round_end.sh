#!/bin/bash
# tools/round_end.sh -- the measurement batch behind profiles/r03_*: run through gpurun from the repo root, then `python tools/traffic_collect.py` here.
#   full -m gpu suite | PMC traffic of every workload (tools/traffic_all.sh) | bench.py (full record) | rocprofv3 --kernel-trace --stats of the bench command | all-ranks multi-GPU step probe
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
python -m pytest tests -m gpu -x -q > gpurun_out/round_end_tests.log 2>&1; tail -2 gpurun_out/round_end_tests.log
bash tools/traffic_all.sh > gpurun_out/round_end_traffic.log 2>&1
# profiles/traffic.json at this build's kernel revision BEFORE the bench runs (the bench attaches an entry only when the revisions match); copies travel back in gpurun_out/
python tools/traffic_collect.py > gpurun_out/round_end_traffic_collect.log 2>&1 && cp profiles/traffic.json gpurun_out/round_end_traffic.json && cp profiles/r03_traffic.md gpurun_out/round_end_traffic.md
python bench.py > gpurun_out/round_end_bench.json.log 2> gpurun_out/round_end_bench.err
export TMPDIR=/tmp
rm -rf gpurun_out/round_end_prof
(cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/round_end_prof -- python3 $GRAFT_REPO_ROOT/bench.py --no-suite --no-vendor --steps 20 > $GRAFT_REPO_ROOT/gpurun_out/round_end_prof.log 2>&1)
PROBE_FULL=1 PROBE_AG_US=0,20,40,60 timeout 1500 python tools/mg_step_probe.py 8 HV15R all > gpurun_out/round_end_mg_allranks.log 2>&1; tail -9 gpurun_out/round_end_mg_allranks.log
# the same with the direct exchange (mgx.hip; loopback: scratch memory of this GPU as the peers, + N us for the links) and the exchange-footprint probes (profiles/r03_exchange_footprint.md)
PROBE_FULL=1 PROBE_EXCHANGE=push PROBE_AG_US=0,40 timeout 1500 python tools/mg_step_probe.py 8 HV15R all > gpurun_out/round_end_mg_allranks_push.log 2>&1; tail -5 gpurun_out/round_end_mg_allranks_push.log
bash tools/fat_exchange_probe.sh > gpurun_out/round_end_fat_exchange.log 2>&1

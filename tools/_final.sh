# final measurement batch of the round (run through gpurun from the repo root)
python -m pytest tests -m gpu -x -q > gpurun_out/r3_final_tests.log 2>&1; tail -2 gpurun_out/r3_final_tests.log
python bench.py > gpurun_out/r3_final_bench.json.log 2> gpurun_out/r3_final_bench.err; tail -c 300 gpurun_out/r3_final_bench.json.log; echo
export TMPDIR=/tmp
(cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/r3_prof_bench -- python3 $GRAFT_REPO_ROOT/bench.py --no-suite --no-vendor --steps 20 > $GRAFT_REPO_ROOT/gpurun_out/r3_prof_bench.log 2>&1)
find gpurun_out/r3_prof_bench -name "*kernel_stats.csv" | head -3
bash tools/traffic_all.sh > gpurun_out/r3_traffic_all.log 2>&1; tail -12 gpurun_out/r3_traffic_all.log
PROBE_FULL=1 PROBE_AG_US=0,20,40,60 timeout 1500 python tools/mg_step_probe.py 8 HV15R all > gpurun_out/r3_mg_allranks_final.log 2>&1; tail -9 gpurun_out/r3_mg_allranks_final.log

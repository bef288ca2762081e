# tools/round_end_r5_bench_only.sh -- the bench + rocprof part of tools/round_end_r5.sh alone (after profiles/traffic.json was regenerated without a new PMC pass)
python3 bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/round_end_bench.json.log 2> gpurun_out/round_end_bench.err
cp gpurun_out/bench_suite.json gpurun_out/round_end_bench_suite.json
export TMPDIR=/tmp
rm -rf gpurun_out/round_end_prof
(cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/round_end_prof -- python3 $GRAFT_REPO_ROOT/bench.py --no-suite --no-vendor --steps 20 > $GRAFT_REPO_ROOT/gpurun_out/round_end_prof.log 2>&1)
cp gpurun_out/bench_suite.json gpurun_out/round_end_prof_suite.json
tail -c 400 gpurun_out/round_end_bench.json.log

python bench.py > gpurun_out/r3_final_bench5.json.log 2> gpurun_out/r3_final_bench5.err; tail -c 150 gpurun_out/r3_final_bench5.json.log; echo
export TMPDIR=/tmp
rm -rf gpurun_out/r3_prof_bench
(cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/r3_prof_bench -- python3 $GRAFT_REPO_ROOT/bench.py --no-suite --no-vendor --steps 20 > $GRAFT_REPO_ROOT/gpurun_out/r3_prof_bench.log 2>&1)
find gpurun_out/r3_prof_bench -name "*kernel_stats.csv" | head -3

# tools/r5_powerlaw_trace.sh -- r5: rocprofv3 --kernel-trace --stats of dasp_bench powerlaw_1M f64: the per-kernel averages behind profiles/r05_long_cb.md (run through gpurun from the repo root)
export TMPDIR=/tmp
cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof_pl_lcb -- $GRAFT_REPO_ROOT/dasp_amd/bin/dasp_bench powerlaw_1M 1 64 50 5 > $GRAFT_REPO_ROOT/gpurun_out/prof_pl_lcb.log 2>&1
cd $GRAFT_REPO_ROOT; python - <<'P'
import csv,glob
for r in list(csv.DictReader(open(glob.glob("gpurun_out/prof_pl_lcb/*/*kernel_stats.csv")[0])))[:8]:
    print(r["Name"][:70], r["Calls"], r["AverageNs"], r["Percentage"])
P
tail -1 gpurun_out/prof_pl_lcb.log | cut -c1-300

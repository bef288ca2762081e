# tools/r5_webbase_pieces.sh -- r5: is webbase-1M f16 bound by the serial chain of its longest row?  long_piece = 0 (one piece per row: no stage-2 launch) against pieces of 2048 / 1024 / 512
# (stage 2 = a second launch), kernel durations from rocprofv3 --kernel-trace --stats
export TMPDIR=/tmp
for lp in 0 2048 1024 512; do
  dasp_amd/bin/dasp_bench webbase-1M 1 16 2000 100 0.75 $lp | sed -E 's/.*pieces=([0-9]+).*\| ([0-9.]+ ms \(event [0-9.]+\)).*graph: ([0-9.]+) ms.*/long_piece='$lp' \2 graph \3/'
  (cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof_wb_lp$lp -- $GRAFT_REPO_ROOT/dasp_amd/bin/dasp_bench webbase-1M 1 16 300 20 0.75 $lp > /dev/null 2>&1)
  python3 - <<P
import csv,glob
for r in list(csv.DictReader(open(glob.glob("gpurun_out/prof_wb_lp$lp/*/*kernel_stats.csv")[0])))[:3]:
    if "dasp_spmv" in r["Name"] or "long_reduce" in r["Name"]: print("   ", r["Name"][:60], r["Calls"], "avg ns", r["AverageNs"])
P
done

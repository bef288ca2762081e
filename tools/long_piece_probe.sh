export TMPDIR=/tmp
for piece in 0 2048 1024 512 256; do
  rm -rf gpurun_out/lp_prof
  (cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/lp_prof -- python3 $GRAFT_REPO_ROOT/tools/long_piece_probe.py $1 $2 $piece 2>&1 | grep long_piece)
  f=$(find gpurun_out/lp_prof -name "*kernel_stats.csv" | head -1)
  grep -E "dasp_spmv|long_reduce" $f | cut -d, -f1-4 | sed 's/EvNS_7DevArgsE//'
done

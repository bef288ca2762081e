# tools/r5_tail_in_ab.sh -- r5: a long piece's partial last chunk inside the stream (PieceSrc) against the build before (tools/ab/libdasp_amd_base2.so)
# (tools/ab/libdasp_amd_base*.so = the library of the commit before the change under test: `bash tools/build_rev.sh <rev> base && mkdir -p tools/ab && cp dasp_amd/variants/base/libdasp_amd.so tools/ab/<name the script uses>`; git-ignored, removed after the run)
export SWEEP_ONLY="long rows,circuit,mixed"
echo "== base"; DASP_AMD_SO=tools/ab/libdasp_amd_base2.so python3 tools/category_sweep.py 2>&1 | grep " us "
echo "== new";  python3 tools/category_sweep.py 2>&1 | grep " us "
for w in "webbase-1M 64" "webbase-1M 16" "powerlaw_1M 64" "powerlaw_1M 64 0.1" "powerlaw_1M 16 0.1" "rmat_2M 64"; do
  for rep in 1 2; do
    DASP_AMD_SO=tools/ab/libdasp_amd_base2.so python3 tools/plan_time.py $w 2>&1 | tail -1
    python3 tools/plan_time.py $w 2>&1 | tail -1
  done
done
timeout 900 python3 -m pytest tests/test_gpu_spmv.py -x -q -m gpu -k "fours or long or extreme or segmented" 2>&1 | grep -E "passed|failed"

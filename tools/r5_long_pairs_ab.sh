# tools/r5_long_pairs_ab.sh -- r5: f64 long pieces with two elements per lane (16-byte value loads, 8-byte id loads) against the build before (tools/ab/libdasp_amd_base4.so)
# (tools/ab/libdasp_amd_base*.so = the library of the commit before the change under test: `bash tools/build_rev.sh <rev> base && mkdir -p tools/ab && cp dasp_amd/variants/base/libdasp_amd.so tools/ab/<name the script uses>`; git-ignored, removed after the run)
export SWEEP_ONLY="long rows,mixed,circuit" SWEEP_PREC=64
echo "== base"; DASP_AMD_SO=tools/ab/libdasp_amd_base4.so python3 tools/category_sweep.py 2>&1 | grep " us "
echo "== new";  python3 tools/category_sweep.py 2>&1 | grep " us "
export DASP_PLACEMENT_TRIALS=4
for w in "rmat_2M 64" "powerlaw_1M 64" "powerlaw_1M 64 0.1" "webbase-1M 64" "HV15R 64" "nlpkkt160 64"; do
  for rep in 1 2; do
    DASP_AMD_SO=tools/ab/libdasp_amd_base4.so python3 tools/plan_time.py $w 2>&1 | tail -1
    python3 tools/plan_time.py $w 2>&1 | tail -1
  done
done
timeout 900 python3 -m pytest tests/test_gpu_spmv.py -x -q -m gpu -k "long or extreme or piece or full_size" 2>&1 | grep -E "passed|failed"

# tools/r5_tail_in_ab2.sh -- r5: the final long-piece form (partial chunk in the stream only where there is one) against the build before (tools/ab/libdasp_amd_base2.so) and the HBM-bound rows once more
# (tools/ab/libdasp_amd_base*.so = the library of the commit before the change under test: `bash tools/build_rev.sh <rev> base && mkdir -p tools/ab && cp dasp_amd/variants/base/libdasp_amd.so tools/ab/<name the script uses>`; git-ignored, removed after the run)
export SWEEP_ONLY="long rows only: all of length 300,mixed"
echo "== base"; DASP_AMD_SO=tools/ab/libdasp_amd_base2.so python3 tools/category_sweep.py 2>&1 | grep " us "
echo "== new";  python3 tools/category_sweep.py 2>&1 | grep " us "
export DASP_PLACEMENT_TRIALS=4
for w in "rmat_2M 64" "powerlaw_1M 64 0.1" "powerlaw_1M 64" "HV15R 64" "Queen_4147 64" "nlpkkt160 64"; do
  for rep in 1 2 3; do
    DASP_AMD_SO=tools/ab/libdasp_amd_base2.so python3 tools/plan_time.py $w 2>&1 | tail -1
    python3 tools/plan_time.py $w 2>&1 | tail -1
  done
done
timeout 1200 python3 -m pytest tests -x -q -m gpu 2>&1 | grep -E "passed|failed"

# tools/r5_slot_shift_ab.sh -- r5: slot_of with shifts instead of an integer division per lane (the 1&3 pairing groups are 8 / 32) against the build before (tools/ab/libdasp_amd_base3.so)
# (tools/ab/libdasp_amd_base*.so = the library of the commit before the change under test: `bash tools/build_rev.sh <rev> base && mkdir -p tools/ab && cp dasp_amd/variants/base/libdasp_amd.so tools/ab/<name the script uses>`; git-ignored, removed after the run)
export SWEEP_ONLY="short rows,circuit,empty rows"
echo "== base"; DASP_AMD_SO=tools/ab/libdasp_amd_base3.so python3 tools/category_sweep.py 2>&1 | grep " us "
echo "== new";  python3 tools/category_sweep.py 2>&1 | grep " us "
export DASP_PLACEMENT_TRIALS=4
for w in "webbase-1M 64" "webbase-1M 16" "powerlaw_1M 64" "rmat_2M 64" "nlpkkt160 64" "HV15R 64"; do
  for rep in 1 2; do
    DASP_AMD_SO=tools/ab/libdasp_amd_base3.so python3 tools/plan_time.py $w 2>&1 | tail -1
    python3 tools/plan_time.py $w 2>&1 | tail -1
  done
done

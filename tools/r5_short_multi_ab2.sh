# tools/r5_short_multi_ab2.sh -- r5: (1) tiles per wave 2 / 4 / 8 on the short-row families; (2) what the tables through the constant address space alone cost or buy on the HBM-bound rows
# (tools/ab/libdasp_amd_base*.so = the library of the commit before the change under test: `bash tools/build_rev.sh <rev> base && mkdir -p tools/ab && cp dasp_amd/variants/base/libdasp_amd.so tools/ab/<name the script uses>`; git-ignored, removed after the run)
# (ktbase = the build before + only that change), four interleaved rounds, fastest of four placements each
export SWEEP_ONLY="short rows" SWEEP_PREC=64
for v in tpw2 "" tpw8; do echo "== tiles per wave: ${v:-4 (product)}"; env ${v:+DASP_AMD_SO=dasp_amd/variants/$v/libdasp_amd.so} python3 tools/category_sweep.py 2>&1 | grep " us "; done
export DASP_PLACEMENT_TRIALS=4
for w in "Queen_4147 64" "powerlaw_1M 64" "HV15R 64"; do
  for rep in 1 2 3 4; do
    DASP_AMD_SO=tools/ab/libdasp_amd_base.so python3 tools/plan_time.py $w 2>&1 | tail -1
    DASP_AMD_SO=dasp_amd/variants/ktbase/libdasp_amd.so python3 tools/plan_time.py $w 2>&1 | tail -1
    python3 tools/plan_time.py $w 2>&1 | tail -1
  done
done

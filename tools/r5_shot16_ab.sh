# tools/r5_shot16_ab.sh -- r5: f16 blocks / pieces of up to 4 steps issued in one shot (kMedShot16 = 4, variant build) against the product's 2
export SWEEP_ONLY="medium rows only,long rows only: all of length 300,mixed,circuit" SWEEP_PREC=16
for v in "" shot16_4m; do echo "== ${v:-product}"; env ${v:+DASP_AMD_SO=dasp_amd/variants/$v/libdasp_amd.so} python3 tools/category_sweep.py 2>&1 | grep " us "; done
for w in "webbase-1M 16" "nlpkkt160 16" "HV15R 16" "cop20k_A 16" "rmat_2M 16 1 two_phase=-1" "powerlaw_1M 16 0.1"; do
  for rep in 1 2; do
    python3 tools/plan_time.py $w 2>&1 | tail -1
    DASP_AMD_SO=dasp_amd/variants/shot16_4m/libdasp_amd.so python3 tools/plan_time.py $w 2>&1 | tail -1
  done
done

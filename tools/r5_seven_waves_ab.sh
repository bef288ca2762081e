# tools/r5_seven_waves_ab.sh -- r5: plans of one-shot blocks on the f64 build held to 7 waves per SIMD (DevicePlan::seven_waves; DASP_SEVEN_WAVES=0 keeps them on the unconstrained build)
export SWEEP_ONLY="medium rows only,circuit,empty rows" SWEEP_PREC=64
echo "== unconstrained build"; DASP_SEVEN_WAVES=0 python3 tools/category_sweep.py 2>&1 | grep " us "
echo "== automatic";  python3 tools/category_sweep.py 2>&1 | grep " us "
export DASP_PLACEMENT_TRIALS=4
for w in "nlpkkt160 64" "nlpkkt160 64 0.1" "webbase-1M 64" "HV15R 64" "Queen_4147 64 0.1"; do
  for rep in 1 2 3; do
    DASP_SEVEN_WAVES=0 python3 tools/plan_time.py $w 2>&1 | tail -1 | sed 's/$/ (unconstrained)/'
    python3 tools/plan_time.py $w 2>&1 | tail -1
  done
done

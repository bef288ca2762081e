# tools/r5_short_seven_waves.sh -- r5: do the short-row paths gain from the 7-waves-per-SIMD f64 build too?  (DASP_SEVEN_WAVES=1 forces it; the automatic choice looks at medium blocks only)
export SWEEP_ONLY="short rows,circuit,empty rows,long rows only: all of length 300" SWEEP_PREC=64
echo "== automatic"; python3 tools/category_sweep.py 2>&1 | grep " us "
echo "== forced 7 waves";  DASP_SEVEN_WAVES=1 python3 tools/category_sweep.py 2>&1 | grep " us "

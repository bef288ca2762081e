# tools/r5_short_ywt.sh -- r5: the written-through y stores (put_y, DevArgs::ywt) on matrices of short rows, where y is a large share of the traffic and a wave's stores are partial lines
export SWEEP_ONLY="short rows,circuit,empty rows" SWEEP_PREC=64
echo "== written through (product)"; python3 tools/category_sweep.py 2>&1 | grep " us "
echo "== plain stores (DASP_Y_WT=0)";  DASP_Y_WT=0 python3 tools/category_sweep.py 2>&1 | grep " us "

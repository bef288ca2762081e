# tools/scratch/tp_blocks_sweep2.sh -- r5: the two-phase column block against the matrix's width, on every family that takes the form
for w in "rmat_2M 16 1" "powerlaw_1M 16 1" "powerlaw_1M 16 0.3" "ljournal-2008 16 0.3" "webbase-1M 16 4" "webbase-1M 16 16" "ljournal-2008-uniform 16 1"; do
  for cb in 8192 16384 32768; do python3 tools/plan_time.py $w tp_col_block=$cb 2>&1 | tail -1; done
done

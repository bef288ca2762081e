# tools/scratch/tp_blocks_sweep.sh -- r5: the two-phase form's block sizes (tp_col_block x tp_row_block) on the two suite graphs that use it
for w in "rmat_2M 16" "ljournal-2008 16"; do
  for cb in 16384 32768 65536; do for rb in 2048 4096 8192; do python3 tools/plan_time.py $w 1 tp_col_block=$cb tp_row_block=$rb 2>&1 | tail -1; done; done
done

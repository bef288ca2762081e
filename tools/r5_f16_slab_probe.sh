# tools/r5_f16_slab_probe.sh -- r5: f16 rows of 5..11 have no regular chunk at all (a 16 x 16 tile is never 75 % full): MFMA blocks of them are all tail steps.  Slabs for exactly those rows
# (slab_max_len=11) on the graph stand-ins, whose rows are not "near" and for which the automatic rule keeps blocks
for w in "webbase-1M 16" "webbase-1M 16 4 two_phase=-1" "ljournal-2008 16 1 two_phase=-1" "rmat_2M 16 1 two_phase=-1" "powerlaw_1M 16 0.1" "webbase-1M-uniform 16"; do
  for s in "" "slab_max_len=11" "slab_max_len=8"; do python3 tools/plan_time.py $w $s 2>&1 | tail -1; done
done

# tools/scratch/y_natural_probe.sh -- r5: what DASP_Y_NATURAL (y in row order instead of the reference's permuted order) costs at full size
export DASP_PLACEMENT_TRIALS=4
for w in "HV15R 64" "nlpkkt160 64" "Queen_4147 64" "HV15R-unstructured 64" "powerlaw_1M 64" "webbase-1M 64" "ljournal-2008 16" "rmat_2M 16"; do
  for o in "" "y_order=1"; do python3 tools/plan_time.py $w 1 $o 2>&1 | tail -1; done
done

# tools/r5_tp_seg_ab.sh -- two-phase plans with 64-element (product) against 32-element segments (variants/seg32: -DDASP_TP_SEG=32)
for w in "ljournal-2008 1 16 500 50" "ljournal-2008-uniform 1 16 500 50"; do
  tools/ab_env.sh "$w" "X=seg64" "LD_PRELOAD=dasp_amd/variants/seg32/libdasp_amd.so"
done

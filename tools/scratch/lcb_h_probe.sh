# tools/scratch/lcb_h_probe.sh -- r5: the column-blocked hub rows' threshold on powerlaw_1M f64: automatic (rows of >= 4096 = 64 per column block) against forced (rows of >= block_longest = 256), and block_longest raised
export DASP_PLACEMENT_TRIALS=4
for o in "" "long_cb=1" "long_cb=1 block_longest=1024" "long_cb=1 block_longest=2048" "long_cb=-1"; do python3 tools/plan_time.py powerlaw_1M 64 1 $o 2>&1 | tail -1; done

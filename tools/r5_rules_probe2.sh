# tools/r5_rules_probe2.sh -- r5: (a) is powerlaw_1M x0.1 f16 bimodal between runs?  (b) column panels / column-blocked hub rows at sizes below the automatic rule's gate
for i in 1 2 3; do python3 tools/plan_time.py powerlaw_1M 16 0.1 2>&1 | tail -1; python3 tools/plan_time.py powerlaw_1M 16 0.1 long_piece=1024 2>&1 | tail -1; done
for s in 0.1 0.3; do
  python3 tools/plan_time.py powerlaw_1M 64 $s 2>&1 | tail -1
  python3 tools/plan_time.py powerlaw_1M 64 $s col_panels=2 2>&1 | tail -1
  python3 tools/plan_time.py powerlaw_1M 64 $s col_panels=3 2>&1 | tail -1
  python3 tools/plan_time.py powerlaw_1M 64 $s col_panels=2 long_cb=-1 2>&1 | tail -1
  python3 tools/plan_time.py powerlaw_1M 16 $s 2>&1 | tail -1
  python3 tools/plan_time.py powerlaw_1M 16 $s two_phase=1 2>&1 | tail -1
done

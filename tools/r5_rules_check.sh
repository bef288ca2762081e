# tools/r5_rules_check.sh -- r5: the three automatic rules changed after tools/size_sweep.py, at the sizes that showed them (auto options only), and powerlaw_1M's panel count beside them
python3 tools/plan_time.py powerlaw_1M 64 0.03 2>&1 | tail -1
python3 tools/plan_time.py powerlaw_1M 16 0.03 2>&1 | tail -1
python3 tools/plan_time.py powerlaw_1M 64 0.3 2>&1 | tail -1
python3 tools/plan_time.py webbase-1M 16 4 2>&1 | tail -1
python3 tools/plan_time.py webbase-1M 64 1 2>&1 | tail -1
python3 tools/plan_time.py powerlaw_1M 64 1 2>&1 | tail -1
python3 tools/plan_time.py powerlaw_1M 64 1 col_panels=2 2>&1 | tail -1
python3 tools/plan_time.py powerlaw_1M 64 1 col_panels=4 2>&1 | tail -1

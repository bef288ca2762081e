# tools/r5_rules_probe.sh -- r5: two automatic rules seen off in tools/size_sweep.py: (1) one piece per long row (no stage 2) when the longest row is a long serial chain in a small matrix;
# (2) the two-phase form on a matrix of very short rows
for s in 0.03 0.1 0.3; do for lp in 0 4096 2048 1024 512; do python3 tools/plan_time.py powerlaw_1M 64 $s long_piece=$lp 2>&1 | tail -1; done; done
for s in 0.03 0.1; do for lp in 0 2048 1024; do python3 tools/plan_time.py powerlaw_1M 16 $s long_piece=$lp 2>&1 | tail -1; done; done
for tp in 0 -1; do python3 tools/plan_time.py webbase-1M 16 16 two_phase=$tp 2>&1 | tail -1; done
for tp in 0 1; do python3 tools/plan_time.py webbase-1M 16 4 two_phase=$tp 2>&1 | tail -1; done
for tp in 0 1; do python3 tools/plan_time.py ljournal-2008 16 0.1 two_phase=$tp 2>&1 | tail -1; python3 tools/plan_time.py rmat_2M 16 0.3 two_phase=$tp 2>&1 | tail -1; done

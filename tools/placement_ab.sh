#!/bin/bash
# tools/placement_ab.sh: the bench headline in fresh processes, placement trials off / on alternating (profiles/r03_placement.md)
for i in 1 2 3 4 5 6; do
  for t in ${PLACEMENT_AB_SET:-1 6}; do
    DASP_PLACEMENT_TRIALS=$t python bench.py --no-suite --no-cpu-baseline --no-vendor --no-random-x --steps 200 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('trials $t:', d['ms_per_step'], d['roofline']['kernel_ms'])"
  done
done

#!/bin/bash
# r6_call8 -- one gpurun batch of round 6 (its output: gpurun_out/r6/; what it measured is quoted in profiles/r06_*.md)
out=gpurun_out/r6; mkdir -p $out
export PYTHONPATH=$PWD
V=$PWD/dasp_amd/variants
timeout 1500 python3 -m pytest tests -m gpu -x -q --deselect tests/test_zz_auto_rules.py > $out/gputest8.log 2>&1; tail -3 $out/gputest8.log
SWEEP_ONLY="long rows,mixed: lengths" timeout 900 python3 tools/category_sweep.py 2>&1 | grep -v amdgpu.ids > $out/longsweep8.log
cat $out/longsweep8.log
timeout 1500 python3 -m pytest tests/test_zz_auto_rules.py -m gpu -x -q > $out/autorules8.log 2>&1; tail -5 $out/autorules8.log
cp gpurun_out/r6_auto_rules.md $out/auto_rules8.md 2>/dev/null
grep LOSS $out/auto_rules8.md
python3 bench.py --gpus 1 --steps 20 --warmup 5 > $out/bench8.json.log 2> $out/bench8.err; tail -c 3000 $out/bench8.json.log; tail -5 $out/bench8.err
cp gpurun_out/bench_suite.json $out/bench8_suite.json

#!/bin/bash
# r6_call11 -- one gpurun batch of round 6 (its output: gpurun_out/r6/; what it measured is quoted in profiles/r06_*.md)
out=gpurun_out/r6; mkdir -p $out
export PYTHONPATH=$PWD
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
timeout 2400 python3 -m pytest tests -m gpu -x -q > $out/gputest11.log 2>&1; tail -5 $out/gputest11.log
cp gpurun_out/r6_auto_rules.md $out/auto_rules11.md 2>/dev/null
grep -E "LOSS|structures" $out/auto_rules11.md
SWEEP_ONLY="short rows only,circuit-like,empty rows" timeout 1200 python3 tools/category_sweep.py 2>&1 | grep -v amdgpu.ids > $out/catsweep11.log; cat $out/catsweep11.log

#!/bin/bash
# r6_call7 -- one gpurun batch of round 6 (its output: gpurun_out/r6/; what it measured is quoted in profiles/r06_*.md)
out=gpurun_out/r6; mkdir -p $out
export PYTHONPATH=$PWD
V=$PWD/dasp_amd/variants
timeout 1500 python3 -m pytest tests -m gpu -x -q --deselect tests/test_zz_auto_rules.py > $out/gputest7.log 2>&1; tail -3 $out/gputest7.log
for v in product r5; do
  if [ $v = product ]; then unset DASP_AMD_SO; else export DASP_AMD_SO=$V/$v/libdasp_amd.so; fi
  echo "== $v"; SWEEP_ONLY="long rows,mixed: lengths,circuit-like" timeout 900 python3 tools/category_sweep.py 2>&1 | grep -v amdgpu.ids
done > $out/longsweep7.log 2>&1
unset DASP_AMD_SO
cat $out/longsweep7.log
timeout 1500 python3 -m pytest tests/test_zz_auto_rules.py -m gpu -x -q > $out/autorules7.log 2>&1; tail -5 $out/autorules7.log
cp gpurun_out/r6_auto_rules.md $out/auto_rules7.md 2>/dev/null

#!/bin/bash
# r6_call10 -- one gpurun batch of round 6 (its output: gpurun_out/r6/; what it measured is quoted in profiles/r06_*.md)
out=gpurun_out/r6; mkdir -p $out
export PYTHONPATH=$PWD
V=$PWD/dasp_amd/variants
timeout 1500 python3 -m pytest tests -m gpu -x -q --deselect tests/test_zz_auto_rules.py > $out/gputest10.log 2>&1; tail -3 $out/gputest9.log
cat > /tmp/r6_suite.py <<'PY'
import sys, os, numpy as np, torch, dasp_amd as D
tag = sys.argv[1]
for name, prec in (("HV15R",64),("cop20k_A",64),("nlpkkt160",64),("powerlaw_1M",64),("Queen_4147",64),("HV15R-unstructured",64),("webbase-1M",16),("ljournal-2008",16),("rmat_2M",16),("webbase-1M",64),("nlpkkt160",16),("HV15R",16)):
    dt, tdt = (np.float64, torch.float64) if prec == 64 else (np.float16, torch.float16)
    rp, ci = D.synth_csr(name, 1.0); m, n = D.synth_dims(name, 1.0)[:2]
    p = D.Plan(rp, ci, np.ones(ci.size, dt), n, precision=prec).upload()
    p.drop_host()
    x = torch.ones(n, dtype=tdt, device="cuda"); y = torch.zeros(m, dtype=tdt, device="cuda")
    it = 1000 if ci.size < 2e7 else 100
    t = [1e3 * p.time(x.data_ptr(), y.data_ptr(), 0, it // 10, it)[1] for _ in range(3)]
    b_alg = ci.size * (prec // 8 + 4) + (m + 1) * 4 + (n + m) * (prec // 8)
    print(tag, name, prec, "us", ["%.2f" % v for v in t], "frac %.3f" % (b_alg / (min(t) * 1e-6) / 8e12), flush=True)
    p.close(); del x, y; torch.cuda.empty_cache()
PY
for v in product r5 product r5; do
  if [ $v = product ]; then unset DASP_AMD_SO; else export DASP_AMD_SO=$V/$v/libdasp_amd.so; fi
  timeout 1200 python3 /tmp/r6_suite.py $v
done > $out/suite_ab10.log 2>&1
unset DASP_AMD_SO
grep -v amdgpu.ids $out/suite_ab10.log | sort -k2,3 -s
for v in product r5; do
  if [ $v = product ]; then unset DASP_AMD_SO; else export DASP_AMD_SO=$V/$v/libdasp_amd.so; fi
  echo "== $v"; SWEEP_ONLY="long rows,mixed: lengths,short rows only" timeout 1200 python3 tools/category_sweep.py 2>&1 | grep -v amdgpu.ids
done > $out/catsweep10.log 2>&1
unset DASP_AMD_SO
cat $out/catsweep10.log
timeout 1500 python3 -m pytest tests/test_zz_auto_rules.py -m gpu -x -q > $out/autorules10.log 2>&1; tail -3 $out/autorules9.log
cp gpurun_out/r6_auto_rules.md $out/auto_rules10.md 2>/dev/null
grep LOSS $out/auto_rules10.md

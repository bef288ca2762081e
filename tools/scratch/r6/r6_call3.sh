#!/bin/bash
# tools/r6_call3.sh -- GPU parity of the r6 window kernels + stamps + A/B against the r5 library on the same box
out=gpurun_out/r6; mkdir -p $out
export PYTHONPATH=$PWD
V=$PWD/dasp_amd/variants
timeout 900 python3 -m pytest tests -m gpu -x -q > $out/gputest3.log 2>&1; tail -3 $out/gputest3.log
{
export DASP_AMD_SO=$V/stamps/libdasp_amd.so
echo "=== cop20k_A f64 auto"; timeout 300 python3 tools/stamp_probe.py cop20k_A 64
echo "=== cop20k_A f64 auto DASP_WIN_FOLD=0"; DASP_WIN_FOLD=0 timeout 300 python3 tools/stamp_probe.py cop20k_A 64 row_window=448
} > $out/stamps3.log 2>&1
unset DASP_AMD_SO
cat > /tmp/r6_time.py <<'PY'
import sys, os, numpy as np, torch, dasp_amd as D
tag = sys.argv[1]
cases = [("cop20k_A",64,1.0,{}),("cop20k_A",64,1.0,{"row_window":448}),("cop20k_A",64,1.0,{"row_window":512}),("cop20k_A",64,0.5,{}),("cop20k_A",64,2.0,{}),("cop20k_A",64,4.0,{}),("cop20k_A",64,16.0,{}),("cop20k_A",16,1.0,{}),("cop20k_A",16,4.0,{})]
if tag == "r5": cases = [c for c in cases if not c[3] or c[3]["row_window"] % 64 == 0]
for name, prec, sc, kw in cases:
    dt, tdt = (np.float64, torch.float64) if prec == 64 else (np.float16, torch.float16)
    rp, ci = D.synth_csr(name, sc); m, n = D.synth_dims(name, sc)[:2]
    p = D.Plan(rp, ci, np.ones(ci.size, dt), n, precision=prec, **kw).upload()
    x = torch.ones(n, dtype=tdt, device="cuda"); y = torch.zeros(m, dtype=tdt, device="cuda")
    t = [1e3 * p.time(x.data_ptr(), y.data_ptr(), 0, 100, 1000)[1] for _ in range(3)]
    want = torch.from_numpy(np.diff(rp).astype(np.float64)[p.order_rid]).cuda()
    ok = bool((y.double() == want).all().item())
    print(tag, name, prec, sc, kw, "windows", p.stats["n_windows"], "R", p.stats["row_window"], "us", ["%.2f" % v for v in t], "exact" if ok else "WRONG", flush=True)
    p.close()
PY
for v in product r5 nt4 nt2 nt1 product r5; do
  if [ $v = product ]; then unset DASP_AMD_SO; else export DASP_AMD_SO=$V/$v/libdasp_amd.so; fi
  timeout 600 python3 /tmp/r6_time.py $v
done > $out/variants3.log 2>&1
unset DASP_AMD_SO
DASP_WIN_FOLD=0 timeout 300 python3 /tmp/r6_time.py product_nofold >> $out/variants3.log 2>&1
cat $out/variants3.log

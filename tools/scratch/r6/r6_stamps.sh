#!/bin/bash
# tools/r6_stamps.sh -- the per-phase stamps of the two small BASELINE matrices (profiles/r06_small_matrix.md) + the window variants on the same box
out=gpurun_out/r6; mkdir -p $out
V=$PWD/dasp_amd/variants
{
export DASP_AMD_SO=$V/stamps/libdasp_amd.so
for rw in 0 448 432; do
  echo "=== cop20k_A f64 row_window=$rw"; timeout 300 python3 tools/stamp_probe.py cop20k_A 64 row_window=$rw
done
echo "=== cop20k_A f64 DASP_WIN1=0"; DASP_WIN1=0 timeout 300 python3 tools/stamp_probe.py cop20k_A 64
echo "=== webbase-1M f16"; timeout 300 python3 tools/stamp_probe.py webbase-1M 16
echo "=== webbase-1M f64"; timeout 300 python3 tools/stamp_probe.py webbase-1M 64
export DASP_AMD_SO=$V/stamps_dyn/libdasp_amd.so
echo "=== DYN cop20k_A f64 row_window=448"; timeout 300 python3 tools/stamp_probe.py cop20k_A 64 row_window=448
} > $out/stamps2.log 2>&1
unset DASP_AMD_SO
cat > /tmp/r6_time.py <<'PY'
import sys, numpy as np, torch, dasp_amd as D
tag = sys.argv[1]
for name, prec, kw in (("cop20k_A",64,{}),("cop20k_A",64,{"row_window":464}),("cop20k_A",64,{"row_window":448}),("cop20k_A",64,{"row_window":432}),("cop20k_A",64,{"row_window":224}),("cop20k_A",64,{"row_window":208})):
    dt, tdt = (np.float64, torch.float64) if prec == 64 else (np.float16, torch.float16)
    rp, ci = D.synth_csr(name, 1.0); m, n = D.synth_dims(name, 1.0)[:2]
    p = D.Plan(rp, ci, np.ones(ci.size, dt), n, precision=prec, **kw).upload()
    x = torch.ones(n, dtype=tdt, device="cuda"); y = torch.zeros(m, dtype=tdt, device="cuda")
    t = [1e3 * p.time(x.data_ptr(), y.data_ptr(), 0, 100, 1000)[1] for _ in range(3)]
    want = torch.from_numpy(np.diff(rp).astype(np.float64)[p.order_rid]).cuda()
    ok = bool((y == want).all().item())
    print(tag, name, prec, kw, "windows", p.stats["n_windows"], "us", ["%.2f" % v for v in t], "exact" if ok else "WRONG", flush=True)
    p.close()
PY
for v in product dyn nt4 nt3 nt2 product; do
  if [ $v = product ]; then unset DASP_AMD_SO; else export DASP_AMD_SO=$V/$v/libdasp_amd.so; fi
  timeout 300 python3 /tmp/r6_time.py $v
done > $out/variants2.log 2>&1
tail -40 $out/variants2.log

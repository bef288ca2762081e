#!/usr/bin/env python3
"""tools/hybrid_probe.py [rows=2000000] [nnz_per_row=14] [band=3000] [far=0.1] [precision=64]: rows of equal length whose columns lie in a
+-band around the row except for a share `far` anywhere -- the strict x windows do not fit (every window spans the whole matrix); compares
the default plan (global gathers) with hybrid windows (x_window_hybrid=1: the band from LDS, the outliers from global memory)."""
import sys
import numpy as np
import torch
sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
import dasp_amd as D

m = int(sys.argv[1]) if len(sys.argv) > 1 else 2_000_000
L = int(sys.argv[2]) if len(sys.argv) > 2 else 14
band = int(sys.argv[3]) if len(sys.argv) > 3 else 3000
far = float(sys.argv[4]) if len(sys.argv) > 4 else 0.1
prec = int(sys.argv[5]) if len(sys.argv) > 5 else 64
rng = np.random.default_rng(3)
rows = np.repeat(np.arange(m), L)
ci = np.where(rng.random(m * L) < far, rng.integers(0, m, m * L), np.clip(rows + rng.integers(-band, band + 1, m * L), 0, m - 1)).astype(np.int32)
rp = (np.arange(m + 1, dtype=np.int64) * L).astype(np.int32)
dt, tdt = (np.float64, torch.float64) if prec == 64 else (np.float16, torch.float16)
v = np.ones(ci.size, dt)
x = torch.ones(m, dtype=tdt, device="cuda")
y = torch.zeros(m, dtype=tdt, device="cuda")
b_alg = ci.size * (prec // 8 + 4) + (m + 1) * 4 + 2 * m * (prec // 8)
for name, kw in (("default", {}), ("hybrid 80K", dict(x_window_hybrid=1)), ("hybrid 160K", dict(x_window_hybrid=1, x_window=163840)), ("no windows", dict(x_window=-1))):
    p = D.Plan(rp, ci, v, m, precision=prec, **kw).upload()
    st = p.stats
    w, e = p.time(x.data_ptr(), y.data_ptr(), 0, warmup=20, iters=200)
    ok = bool((y.double() == L).all().item())
    print("%-12s windows=%d hybrid=%d lds_share=%.2f | %.4f ms  %.3f of 8 TB/s  ok=%s" % (name, st["x_window_on"], st["x_window_hybrid"], st["window_nnz_frac"], e, b_alg / (e * 1e6) / 8000, ok))
    p.close()

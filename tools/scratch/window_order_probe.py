"""tools/scratch/window_order_probe.py -- r5: rows of mixed lengths with local columns: the global length sort puts 16 rows from anywhere into a block (16 regions of x per chunk).  The windowed order
(sort inside windows of row_window rows) without and with LDS staging against the default plan"""
import os, sys, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import dasp_amd as D
src = open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'category_sweep.py')).read().split("FAMILIES = [")[0]
exec(src[src.index("rng = "):])
M = 1 << 20
cases = [("5..255 local", from_lengths(rng.integers(5, 256, M), M, 512), M), ("10..60 local (FEM-like)", from_lengths(rng.integers(10, 61, 4 * M), 4 * M, 512), 4 * M),
         ("mixed 0..600", from_lengths(rng.integers(0, 601, 300000), 4 * M, 2048), 4 * M)]
for name in ("HV15R-unstructured",):
    rp, ci = D.synth_csr(name, 1.0); cases.append((name, (rp, ci), D.synth_dims(name, 1.0)[1]))
for desc, (rp, ci), n in cases:
    m = rp.size - 1
    for prec in (64, 16):
        for kw in ({}, dict(x_window=-2, row_window=1024), dict(x_window=-2, row_window=256), dict(x_window=81920, row_window=1024), dict(x_window=163840, row_window=512)):
            dt = np.float64 if prec == 64 else np.float16
            try:
                plan = D.Plan(rp, ci, np.ones(ci.size, dt), n, precision=prec, **kw).upload()
            except Exception as e:
                print(desc, prec, kw, "ERROR", str(e)[:80]); continue
            plan.drop_host()
            tdt = torch.float64 if prec == 64 else torch.float16
            x = torch.ones(n, dtype=tdt, device="cuda"); y = torch.zeros(m, dtype=tdt, device="cuda")
            best = min(plan.time(x.data_ptr(), y.data_ptr(), 0, warmup=20, iters=100)[1] for _ in range(3))
            b_alg = ci.size * (prec // 8 + 4) + (m + 1) * 4 + (n + m) * (prec // 8)
            st = plan.stats
            print("%-26s f%d %-42s %9.1f us %.3f  windows %d (LDS %d) frac %.2f" % (desc, prec, kw, best * 1e3, b_alg / (best * 1e6) / 8000, st["n_windows"], st["n_windows_lds"], st["window_nnz_frac"]), flush=True)
            plan.close(); del x, y, plan; torch.cuda.empty_cache()

"""tools/scratch/window_size_probe.py -- r5: from which size on do LDS windows pay on rows of mixed lengths with local columns (the global length sort scatters a block's 16 rows; windows sort inside 1024 rows)?"""
import os, sys, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import dasp_amd as D
src = open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'category_sweep.py')).read().split("FAMILIES = [")[0]
exec(src[src.index("rng = "):])
M = 1 << 20
for desc, lo, hi, rows_list in (("5..255", 5, 256, (M // 64, M // 16, M // 4, M)), ("10..60", 10, 61, (M // 16, M // 4, M, 4 * M))):
    for rows in rows_list:
        rp, ci = from_lengths(rng.integers(lo, hi, rows), rows, 512)
        m = n = rows
        for prec in (64, 16):
            out = []
            for kw in ({}, dict(x_window=81920), dict(x_window=163840)):
                dt = np.float64 if prec == 64 else np.float16
                plan = D.Plan(rp, ci, np.ones(ci.size, dt), n, precision=prec, **kw).upload(); plan.drop_host()
                tdt = torch.float64 if prec == 64 else torch.float16
                x = torch.ones(n, dtype=tdt, device="cuda"); y = torch.zeros(m, dtype=tdt, device="cuda")
                best = min(plan.time(x.data_ptr(), y.data_ptr(), 0, warmup=20, iters=200)[1] for _ in range(3))
                st = plan.stats
                out.append("%8.1f us (%d windows of %d)" % (best * 1e3, st["n_windows_lds"], st["row_window"]))
                plan.close(); del x, y, plan; torch.cuda.empty_cache()
            print("%-7s rows %8d nnz %9d f%d | plain %s | 80 KiB %s | 160 KiB %s" % (desc, rows, ci.size, prec, out[0], out[1], out[2]), flush=True)

"""tools/scratch/circ_probe.py -- r5: rows of 5..8 (a circuit-like mix, 8 M local rows of 5..8, 8 M stencil-like rows of 7) as MFMA blocks (slab_max_len=4), as slabs (16) and under the automatic rule, f64 and f16: the f16 rows have no regular chunk at all (profiles/r05_category_sweep.md section 5)"""
import os, sys, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import dasp_amd as D
src = open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'category_sweep.py')).read().split("FAMILIES = [")[0]
exec(src[src.index("rng = "):])
M = 1 << 20
for desc, (rp, ci), n in (("circuit", from_lengths(np.concatenate([rng.integers(1, 9, 4 * M - 60), np.full(60, 100000)])[rng.permutation(4 * M)], 4 * M, 256), 4 * M),
                          ("rows 5..8", from_lengths(rng.integers(5, 9, 8 * M), 8 * M, 256), 8 * M),
                          ("rows of 7 (stencil-like, local)", from_lengths(np.full(8 * M, 7), 8 * M, 4), 8 * M)):
    m = rp.size - 1
    for prec in (64, 16):
        for kw in ({}, dict(slab_max_len=16), dict(slab_max_len=4)):
            dt = np.float64 if prec == 64 else np.float16
            plan = D.Plan(rp, ci, np.ones(ci.size, dt), n, precision=prec, **kw).upload(); plan.drop_host()
            tdt = torch.float64 if prec == 64 else torch.float16
            x = torch.ones(n, dtype=tdt, device="cuda"); y = torch.zeros(m, dtype=tdt, device="cuda")
            best = min(plan.time(x.data_ptr(), y.data_ptr(), 0, warmup=20, iters=100)[1] for _ in range(3))
            b_alg = ci.size * (prec // 8 + 4) + (m + 1) * 4 + (n + m) * (prec // 8)
            st = plan.stats
            print("%-32s f%d %-22s %8.1f us %.3f blocks %d tiles %d" % (desc, prec, kw, best * 1e3, b_alg / (best * 1e6) / 8000, st["n_med_blocks"], st["n_short_tiles"]), flush=True)
            plan.close(); del x, y, plan; torch.cuda.empty_cache()

"""tools/scratch/thr_probe.py -- r5 probe: the regular / irregular split threshold (reference default 0.75 of a tile) on f16, whose tiles are 16 x 16: more zero fill, fewer tail steps"""
import os, sys, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import dasp_amd as D
src = open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'category_sweep.py')).read().split("FAMILIES = [")[0]
exec(src[src.index("rng = "):])
M = 1 << 20
cases = [("len 40", from_lengths(np.full(2 * M, 40), 2 * M, 256), 2 * M), ("len 17", from_lengths(np.full(4 * M, 17), 4 * M, 256), 4 * M), ("5..255", from_lengths(rng.integers(5, 256, M), M, 512), M)]
for name, scale in (("nlpkkt160", 1.0), ("HV15R", 1.0), ("webbase-1M", 1.0), ("cop20k_A", 1.0)):
    rp, ci = D.synth_csr(name, scale); cases.append((name, (rp, ci), D.synth_dims(name, scale)[1]))
for desc, (rp, ci), n in cases:
    m = rp.size - 1
    for prec in (16, 64):
        for thr in (0.75, 0.5, 0.3):
            dt = np.float64 if prec == 64 else np.float16
            plan = D.Plan(rp, ci, np.ones(ci.size, dt), n, precision=prec, threshold=thr).upload(); plan.drop_host()
            tdt = torch.float64 if prec == 64 else torch.float16
            x = torch.ones(n, dtype=tdt, device="cuda"); y = torch.zeros(m, dtype=tdt, device="cuda")
            best = min(plan.time(x.data_ptr(), y.data_ptr(), 0, warmup=20, iters=100)[1] for _ in range(3))
            b_alg = ci.size * (prec // 8 + 4) + (m + 1) * 4 + (n + m) * (prec // 8)
            st = plan.stats
            print("%-12s f%d threshold %.2f %9.1f us %.3f  fill0 %.3f irregular %d" % (desc, prec, thr, best * 1e3, b_alg / (best * 1e6) / 8000, st["rate_fill0"], st["nnz_irreg"]), flush=True)
            plan.close(); del x, y, plan; torch.cuda.empty_cache()

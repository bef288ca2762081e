"""tools/scratch/shape_probe.py -- r5: extreme shapes at scale under the automatic plan: one huge row, a few huge rows, tall-skinny short rows over a tiny x (the gathers all hit the L1: the short-row path's
streaming rate without its gather), wide-short"""
import os, sys, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import dasp_amd as D
rng = np.random.default_rng(21)
M = 1 << 20
def rows_of(lens, n, sort=True):
    m = lens.size
    rp = np.zeros(m + 1, np.int64); np.cumsum(lens, out=rp[1:])
    ci = rng.integers(0, n, int(rp[-1]), dtype=np.int64)
    if sort:
        rows = np.repeat(np.arange(m, dtype=np.int64), lens)
        o = np.lexsort((ci, rows)); ci = ci[o]
    return rp.astype(np.int32), ci.astype(np.int32)
cases = [("one row of 50 M (columns 0..50 M)", (np.array([0, 50 * M], np.int32), np.arange(50 * M, dtype=np.int32)), 50 * M),
         ("ten rows of 8 M, consecutive columns", (np.arange(0, 11 * 8 * M, 8 * M).astype(np.int32), np.tile(np.arange(8 * M, dtype=np.int32), 10)), 8 * M),
         ("tall-skinny: 48 M rows of 3 over 1000 columns", rows_of(np.full(48 * M, 3), 1000, sort=False), 1000),
         ("tall-skinny: 48 M rows of 1 over 1000 columns", rows_of(np.full(48 * M, 1), 1000, sort=False), 1000),
         ("tall-skinny: 4 M rows of 30 over 4096 columns", rows_of(np.full(4 * M, 30), 4096, sort=False), 4096)]
for desc, (rp, ci), n in cases:
    m = rp.size - 1
    for prec in (64, 16):
        dt = np.float64 if prec == 64 else np.float16
        vals = np.full(ci.size, 1.0 / 1024 if m < 100 else 1.0, dt)
        plan = D.Plan(rp, ci, vals, n, precision=prec).upload(); plan.drop_host()
        tdt = torch.float64 if prec == 64 else torch.float16
        x = torch.ones(n, dtype=tdt, device="cuda"); y = torch.zeros(m, dtype=tdt, device="cuda")
        best = min(plan.time(x.data_ptr(), y.data_ptr(), 0, warmup=10, iters=50)[1] for _ in range(3))
        want = torch.from_numpy((np.diff(rp)[plan.order_rid].astype(np.float64)) * float(vals[0])).cuda()
        ok = bool(((y.double() - want).abs() <= (1e-12 if prec == 64 else 1e-2) * want.clamp(min=1)).all().item())
        b_alg = ci.size * (prec // 8 + 4) + (m + 1) * 4 + (n + m) * (prec // 8)
        st = plan.stats
        print("%-52s f%d nnz %10d %8.1f MB %9.1f us %.3f pieces %d multi %d blocks %d short tiles %d %s" % (desc, prec, ci.size, b_alg / 1e6, best * 1e3, b_alg / (best * 1e6) / 8000, st["n_long_pieces"], st["n_long_multi"], st["n_med_blocks"], st["n_short_tiles"], "ok" if ok else "WRONG"), flush=True)
        plan.close(); del x, y, plan; torch.cuda.empty_cache()

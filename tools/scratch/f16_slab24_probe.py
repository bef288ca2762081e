"""tools/scratch/f16_slab24_probe.py -- r5: f16 rows of 12..24 nonzeros whose equally long neighbours are close but not within a line: MFMA blocks (automatic), slabs (slab_max_len=24), LDS windows (forced)"""
import os, sys, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import dasp_amd as D
src = open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'category_sweep.py')).read().split("FAMILIES = [")[0]
exec(src[src.index("rng = "):])
M = 1 << 20
def band(m, per, half, outliers):
    rows = np.repeat(np.arange(m, dtype=np.int64), per)
    ci = np.where(rng.random(rows.size) < 1 - outliers, np.clip(rows + rng.integers(-half, half + 1, rows.size), 0, m - 1), rng.integers(0, m, rows.size))
    o = np.lexsort((ci, rows))
    return (np.arange(m + 1, dtype=np.int64) * per).astype(np.int32), ci[o].astype(np.int32)
cases = [("rows of 12, band +-500, 10 % anywhere (2 M rows)",) + band(2 * M, 12, 500, 0.1) + (2 * M,),
         ("rows of 14, band +-3000, 10 % anywhere (2 M rows)",) + band(2 * M, 14, 3000, 0.1) + (2 * M,),
         ("rows of 17, runs, starts +-256 (4 M rows)",) + from_lengths(np.full(4 * M, 17), 4 * M, 256) + (4 * M,),
         ("rows of 12..24, runs, starts +-256 (4 M rows)",) + from_lengths(rng.integers(12, 25, 4 * M), 4 * M, 256) + (4 * M,),
         ("rows of 24, runs, starts +-64 (3 M rows)",) + from_lengths(np.full(3 * M, 24), 3 * M, 64) + (3 * M,)]
for desc, rp, ci, n in cases:
    m = rp.size - 1
    res = []
    for kw in ({}, dict(slab_max_len=24), dict(slab_max_len=4, x_window=-1), dict(x_window=81920), dict(x_window=81920, x_window_hybrid=1)):
        try:
            plan = D.Plan(rp, ci, np.ones(ci.size, np.float16), n, precision=16, **kw).upload()
        except Exception as e:
            res.append("%s error" % kw); continue
        plan.drop_host()
        x = torch.ones(n, dtype=torch.float16, device="cuda"); y = torch.zeros(m, dtype=torch.float16, device="cuda")
        best = min(plan.time(x.data_ptr(), y.data_ptr(), 0, warmup=20, iters=100)[1] for _ in range(3))
        b_alg = ci.size * 6 + (m + 1) * 4 + (n + m) * 2
        st = plan.stats
        form = ("win%s" % ("h" if st["x_window_hybrid"] else "")) if st["x_window_on"] else ("slab" if st["n_med_blocks"] == 0 and st["n_short_tiles"] > 0 else "blk")
        res.append("%s %.3f(%s)" % (",".join("%s=%s" % kv for kv in kw.items()) or "auto", b_alg / (best * 1e6) / 8000, form))
        plan.close(); del x, y, plan; torch.cuda.empty_cache()
    print("%-52s nnz %9d | %s" % (desc, ci.size, " | ".join(res)), flush=True)

"""tools/scratch/cid8_oneshot_probe.py -- r5: one-byte ids in ONE-SHOT blocks (rows of <= 32 nonzeros: pairs of chunks, 16-bit id loads per lane) against 16-bit ids (cid8=-1) on stencil-like matrices"""
import os, sys, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import dasp_amd as D
src = open(os.path.join(os.path.dirname(os.path.abspath(__file__)), 'structure_probe.py')).read()
exec(src[src.index("src = open"):src.index("g = 160")])
M = 1 << 20
g = 160
offs27 = [dz * g * g + dy * g + dx for dz in (-1, 0, 1) for dy in (-1, 0, 1) for dx in (-1, 0, 1)]
cases = [("27-point stencil 160^3",) + stencil(g ** 3, offs27) + (g ** 3,)]
lens = np.where(rng.random(2 * M) < 0.5, 30, 45); cases.append(("two lengths 30 / 45 local",) + from_lengths(lens, 2 * M, 128) + (2 * M,))
cases.append(("rows of 17 local",) + from_lengths(np.full(4 * M, 17), 4 * M, 256) + (4 * M,))
cases.append(("rows of 28 local",) + from_lengths(np.full(3 * M, 28), 3 * M, 256) + (3 * M,))
cases.append(("rows of 40 local (pipelined)",) + from_lengths(np.full(2 * M, 40), 2 * M, 256) + (2 * M,))
for desc, rp, ci, n in cases:
    m = rp.size - 1
    for kw in ({}, dict(cid8=-1)):
        plan = D.Plan(rp, ci, np.ones(ci.size), n, precision=64, **kw).upload(); plan.drop_host()
        x = torch.ones(n, dtype=torch.float64, device="cuda"); y = torch.zeros(m, dtype=torch.float64, device="cuda")
        best = min(plan.time(x.data_ptr(), y.data_ptr(), 0, warmup=20, iters=100)[1] for _ in range(3))
        b_alg = ci.size * 12 + (m + 1) * 4 + (n + m) * 8
        st = plan.stats
        print("%-30s %-12s %9.1f us %.3f  narrow chunks %d of %d, %.2f B/nnz packed" % (desc, kw, best * 1e3, b_alg / (best * 1e6) / 8000, st["cid8_chunks"], st["n_med_blocks"], st["data_X"] / ci.size), flush=True)
        plan.close(); del x, y, plan; torch.cuda.empty_cache()

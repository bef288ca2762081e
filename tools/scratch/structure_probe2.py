"""tools/scratch/structure_probe2.py -- r5: more f64 / f16 structures under the automatic plan AND under the options that force another form: is any automatic choice off by more than a few per cent?"""
import os, sys, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import dasp_amd as D
src = open(os.path.join(os.path.dirname(os.path.abspath(__file__)), 'structure_probe.py')).read()
exec(src[src.index("src = open"):src.index("g = 160")])
M = 1 << 20
cases = []
g2 = 4096
cases.append(("9-point stencil 4096^2",) + stencil(g2 * g2, [dy * g2 + dx for dy in (-1, 0, 1) for dx in (-1, 0, 1)]) + (g2 * g2,))
cases.append(("band of half-width 50 (rows of 101)",) + stencil(M, list(range(-50, 51))) + (M,))
rp, ci = from_lengths(np.clip(rng.normal(15, 4, 6 * M).astype(np.int64), 6, 30), 6 * M, 64); cases.append(("tetra-like: lengths ~N(15,4) in 6..30, local",) + (rp, ci, 6 * M))
gq = 110
offs81 = sorted(3 * (dz * gq * gq + dy * gq + dx) + d for dz in (-1, 0, 1) for dy in (-1, 0, 1) for dx in (-1, 0, 1) for d in (0, 1, 2))
cases.append(("27-point stencil x 3 dof (rows of 81) on 110^3",) + stencil(3 * gq ** 3, offs81) + (3 * gq ** 3,))
variants = [{}, dict(cid8=-1), dict(cid8=1), dict(x_window=81920), dict(slab_max_len=32), dict(slab_max_len=4), dict(chunk_pairs=1), dict(chunk_pairs=2)]
for desc, rp, ci, n in cases:
    m = rp.size - 1
    for prec in (64, 16):
        res = []
        for kw in variants:
            dt = np.float64 if prec == 64 else np.float16
            try:
                plan = D.Plan(rp, ci, np.ones(ci.size, dt), n, precision=prec, **kw).upload()
            except Exception as e:
                res.append("%s: error" % kw); continue
            plan.drop_host()
            tdt = torch.float64 if prec == 64 else torch.float16
            x = torch.ones(n, dtype=tdt, device="cuda"); y = torch.zeros(m, dtype=tdt, device="cuda")
            best = min(plan.time(x.data_ptr(), y.data_ptr(), 0, warmup=20, iters=100)[1] for _ in range(3))
            b_alg = ci.size * (prec // 8 + 4) + (m + 1) * 4 + (n + m) * (prec // 8)
            st = plan.stats
            form = "win" if st["x_window_on"] else ("slab" if st["n_med_blocks"] == 0 and st["n_short_tiles"] > 0 else "blk")
            res.append("%s %.3f(%s%s)" % (",".join("%s=%s" % kv for kv in kw.items()) or "auto", b_alg / (best * 1e6) / 8000, form, ",c8" if st["cid8_chunks"] else ""))
            plan.close(); del x, y, plan; torch.cuda.empty_cache()
        print("%-48s f%d nnz %9d | %s" % (desc, prec, ci.size, " | ".join(res)), flush=True)

"""tools/scratch/structure_probe.py -- r5: a few more structures beside the suite and the category sweep, automatic plans only: what form does the library pick and where does it land?"""
import os, sys, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import dasp_amd as D
src = open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'category_sweep.py')).read().split("FAMILIES = [")[0]
exec(src[src.index("rng = "):])
M = 1 << 20

def stencil(m, offs, drop=0.0):
    """row r has columns r + offs (clipped), a fraction `drop` of the entries removed at random (boundary-like rows of other lengths)"""
    offs = np.asarray(offs, np.int64)
    cols = np.arange(m, dtype=np.int64)[:, None] + offs[None, :]
    keep = (cols >= 0) & (cols < m)
    if drop: keep &= rng.random(cols.shape) >= drop
    lens = keep.sum(1)
    rp = np.zeros(m + 1, np.int64); np.cumsum(lens, out=rp[1:])
    return rp.astype(np.int32), cols[keep].astype(np.int32)

def permuted(rp, ci, m):
    """the same matrix with rows AND columns renumbered at random (a mesh without a bandwidth-reducing ordering)"""
    perm = rng.permutation(m)
    inv = np.empty(m, np.int64); inv[perm] = np.arange(m)
    lens = np.diff(rp)[perm]
    rp2 = np.zeros(m + 1, np.int64); np.cumsum(lens, out=rp2[1:])
    idx = np.concatenate([np.arange(rp[r], rp[r + 1]) for r in perm[:0]]) if False else None
    starts = rp[:-1][perm]
    take = np.repeat(starts - rp2[:-1], lens) + np.arange(int(rp2[-1]))
    return rp2.astype(np.int32), inv[ci[take]].astype(np.int32)

g = 160
offs27 = [dz * g * g + dy * g + dx for dz in (-1, 0, 1) for dy in (-1, 0, 1) for dx in (-1, 0, 1)]
cases = []
rp, ci = stencil(g ** 3, offs27); cases.append(("27-point stencil, 160^3, natural order", rp, ci, g ** 3))
rp, ci = stencil(g ** 3, offs27, drop=0.15); cases.append(("the same with 15 % of the entries dropped (many row lengths)", rp, ci, g ** 3))
rp7, ci7 = stencil(256 ** 3, [-256 * 256, -256, -1, 0, 1, 256, 256 * 256]); cases.append(("7-point stencil, 256^3", rp7, ci7, 256 ** 3))
rp, ci = stencil(2 * M, offs27[:27], drop=0.0); rp, ci = permuted(rp, ci, 2 * M); cases.append(("27-point stencil on 2 M points, rows and columns renumbered at random", rp, ci, 2 * M))
lens = np.where(rng.random(2 * M) < 0.5, 30, 45); rp, ci = from_lengths(lens, 2 * M, 128); cases.append(("two row lengths (30 / 45) at random, local", rp, ci, 2 * M))
for desc, rp, ci, n in cases:
    m = rp.size - 1
    for prec in (64, 16):
        dt = np.float64 if prec == 64 else np.float16
        plan = D.Plan(rp, ci, np.ones(ci.size, dt), n, precision=prec).upload(); plan.drop_host()
        tdt = torch.float64 if prec == 64 else torch.float16
        x = torch.ones(n, dtype=tdt, device="cuda"); y = torch.zeros(m, dtype=tdt, device="cuda")
        best = min(plan.time(x.data_ptr(), y.data_ptr(), 0, warmup=20, iters=100)[1] for _ in range(3))
        want = torch.from_numpy(np.diff(rp)[plan.order_rid].astype(np.float64)).cuda()
        ok = bool(((y.double() - want).abs() <= (0.0 if prec == 64 else 1e-2) * want.clamp(min=1)).all().item())
        b_alg = ci.size * (prec // 8 + 4) + (m + 1) * 4 + (n + m) * (prec // 8)
        st = plan.stats
        form = "two-phase" if st["two_phase"] else ("%d panels" % st["n_col_panels"] if st["n_col_panels"] else ("LDS windows" if st["x_window_on"] else "plain"))
        print("%-78s f%d nnz %10d %8.1f MB %9.1f us %.3f %s blocks %d short tiles %d %s" % (desc, prec, ci.size, b_alg / 1e6, best * 1e3, b_alg / (best * 1e6) / 8000, form, st["n_med_blocks"], st["n_short_tiles"], "exact" if ok else "WRONG"), flush=True)
        plan.close(); del x, y, plan; torch.cuda.empty_cache()

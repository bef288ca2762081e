#!/usr/bin/env python3
"""tools/category_knobs.py -- r5: the weak rows of tools/category_sweep.py under the options that choose another storage / kernel for the same rows"""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import dasp_amd as D
src = open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "category_sweep.py")).read().split("FAMILIES = [")[0]
exec(src[src.index("rng = "):])
M = 1 << 20
CASES = [
    ("short 1..4", lambda: from_lengths(rng.integers(1, 5, 24 * M), 24 * M, 64), 24 * M, [(64, {}), (64, dict(short_seg=-1)), (16, {}), (16, dict(short_seg=1))]),
    ("short len 1", lambda: from_lengths(np.full(48 * M, 1), 48 * M, 64), 48 * M, [(64, {}), (64, dict(short_seg=-1))]),
    ("medium len 17", lambda: from_lengths(np.full(4 * M, 17), 4 * M, 256), 4 * M, [(64, {}), (64, dict(slab_max_len=32)), (64, dict(slab_max_len=4)), (16, {}), (16, dict(slab_max_len=32)), (16, dict(slab_max_len=4)), (16, dict(chunk_pairs=-1))]),
    ("medium len 40", lambda: from_lengths(np.full(2 * M, 40), 2 * M, 256), 2 * M, [(64, {}), (16, {}), (16, dict(chunk_pairs=-1)), (16, dict(cid16=-1))]),
    ("long len 300", lambda: from_lengths(np.full(300000, 300), 4 * M, 4096), 4 * M, [(64, {}), (64, dict(block_longest=512)), (16, {}), (16, dict(block_longest=512))]),
    ("long len 2000", lambda: from_lengths(np.full(40000, 2000), 4 * M, 4096), 4 * M, [(64, {}), (64, dict(long_piece=512)), (64, dict(long_piece=2048)), (64, dict(block_longest=4096)), (16, {}), (16, dict(long_piece=2048)), (16, dict(block_longest=4096))]),
    ("long 400 x 200000", lambda: from_lengths(np.full(400, 200000), 4 * M, 4096), 4 * M, [(64, {}), (64, dict(long_piece=4096)), (64, dict(long_piece=16384)), (16, {}), (16, dict(long_piece=4096)), (16, dict(long_piece=16384))]),
]
for desc, make, n, variants in CASES:
    rp, ci = make()
    m = rp.size - 1
    for prec, kw in variants:
        dt = np.float64 if prec == 64 else np.float16
        try:
            plan = D.Plan(rp, ci, np.ones(ci.size, dt), n, precision=prec, **kw).upload()
        except Exception as e:
            print(desc, prec, kw, "ERROR", str(e)[:100]); continue
        plan.drop_host()
        tdt = torch.float64 if prec == 64 else torch.float16
        x = torch.ones(n, dtype=tdt, device="cuda"); y = torch.zeros(m, dtype=tdt, device="cuda")
        best = min(plan.time(x.data_ptr(), y.data_ptr(), 0, warmup=20, iters=100)[1] for _ in range(3))
        want = torch.from_numpy(np.diff(rp)[plan.order_rid].astype(np.float64)).cuda()
        ok = bool(((y.double() - want).abs() <= (0.0 if prec == 64 else 1e-2) * want.clamp(min=1)).all().item())
        b_alg = ci.size * (prec // 8 + 4) + (m + 1) * 4 + (n + m) * (prec // 8)
        st = plan.stats
        print("%-18s f%d %-28s %9.1f us  %.3f  long/medium rows %d/%d pieces %d blocks %d short tiles %d %s" % (desc, prec, kw, best * 1e3, b_alg / (best * 1e6) / 8000, st["row_long"], st["row_block"], st["n_long_pieces"], st["n_med_blocks"], st["n_short_tiles"], "exact" if ok else "WRONG"), flush=True)
        plan.close(); del x, y, plan
        torch.cuda.empty_cache()

#!/usr/bin/env python3
"""tools/category_sweep.py [out.md] -- r5: each row category's kernel at HBM scale on matrices built to hold (almost) only that category -- the suite's stand-ins are
dominated by medium rows.  f64 and f16; fraction of 8 TB/s over the CSR bytes; all-ones exact check (f16: rows up to 2048).  Columns are local (a band around the row) unless
the family says otherwise, so that the streamed matrix, not the x gather, is what is measured."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import dasp_amd as D

rng = np.random.default_rng(11)

def from_lengths(lens, n, band):
    """rows with the given lengths; row r's columns are consecutive from a start inside +-band of r * n / m (ascending, distinct)"""
    m = lens.size
    rp = np.zeros(m + 1, np.int64); np.cumsum(lens, out=rp[1:])
    rows = np.repeat(np.arange(m, dtype=np.int64), lens)
    k = np.arange(int(rp[-1]), dtype=np.int64) - rp[rows]
    centre = rows * n // max(m, 1)
    start = np.clip(centre + rng.integers(-band, band + 1, m)[rows] - lens[rows] // 2, 0, np.maximum(n - lens[rows], 0))
    return rp.astype(np.int32), (start + k).astype(np.int32)

def strided(lens, n, stride):
    """the same, columns `stride` apart (every entry its own 128-byte line of x for stride >= 16 f64 / 64 f16)"""
    m = lens.size
    rp = np.zeros(m + 1, np.int64); np.cumsum(lens, out=rp[1:])
    rows = np.repeat(np.arange(m, dtype=np.int64), lens)
    k = np.arange(int(rp[-1]), dtype=np.int64) - rp[rows]
    start = rng.integers(0, stride, m)[rows]
    return rp.astype(np.int32), ((start + k * stride) % n).astype(np.int32)

M = 1 << 20
FAMILIES = [
    ("short rows only: 1..4 nonzeros (road-network like), local", lambda: from_lengths(rng.integers(1, 5, 24 * M), 24 * M, 64), 24 * M),
    ("short rows only: all of length 3", lambda: from_lengths(np.full(24 * M, 3), 24 * M, 64), 24 * M),
    ("short rows only: all of length 1 (a permuted diagonal, local)", lambda: from_lengths(np.full(48 * M, 1), 48 * M, 64), 48 * M),
    ("medium rows only: all of length 40, local", lambda: from_lengths(np.full(2 * M, 40), 2 * M, 256), 2 * M),
    ("medium rows only: lengths 5..255 uniform, local", lambda: from_lengths(rng.integers(5, 256, M), M, 512), M),
    ("medium rows only: all of length 17 (one chunk + a tail step), local", lambda: from_lengths(np.full(4 * M, 17), 4 * M, 256), 4 * M),
    ("long rows only: all of length 2000, local", lambda: from_lengths(np.full(40000, 2000), 4 * M, 4096), 4 * M),
    ("long rows only: all of length 300, local", lambda: from_lengths(np.full(300000, 300), 4 * M, 4096), 4 * M),
    ("long rows only: 400 rows of 200 000", lambda: from_lengths(np.full(400, 200000), 4 * M, 4096), 4 * M),
    ("circuit-like: 4 M rows of 1..8 and 60 rows of 100 000", lambda: from_lengths(np.concatenate([rng.integers(1, 9, 4 * M - 60), np.full(60, 100000)])[rng.permutation(4 * M)], 4 * M, 256), 4 * M),
    ("empty rows: 3 of 4 rows empty, the others of length 30", lambda: from_lengths(np.where(rng.random(8 * M) < 0.25, 30, 0), 8 * M, 256), 8 * M),
    ("mixed: lengths 0..600 uniform", lambda: from_lengths(rng.integers(0, 601, 300000), 4 * M, 2048), 4 * M),
]
if os.environ.get("SWEEP_ONLY"):        # comma-separated substrings of the descriptions
    FAMILIES = [f for f in FAMILIES if any(k in f[0] for k in os.environ["SWEEP_ONLY"].split(","))]
if os.environ.get("SWEEP_QUICK"):
    FAMILIES = [(d, (lambda f=f: (lambda rp, ci: (rp[:20001], ci[:rp[20000]]))(*f())), n) for d, f, n in FAMILIES[:3]]
rows_out = []
for desc, make, n in FAMILIES:
    rp, ci = make()
    m = rp.size - 1
    for prec in [int(p) for p in os.environ.get("SWEEP_PREC", "64,16").split(",")]:
        t0 = time.time()
        dt = np.float64 if prec == 64 else np.float16
        plan = D.Plan(rp, ci, np.ones(ci.size, dt), n, precision=prec).upload()
        plan.drop_host()
        tdt = torch.float64 if prec == 64 else torch.float16
        x = torch.ones(n, dtype=tdt, device="cuda"); y = torch.zeros(m, dtype=tdt, device="cuda")
        best = 1e9
        for rep in range(3):
            w, e = plan.time(x.data_ptr(), y.data_ptr(), 0, warmup=20, iters=100)
            best = min(best, e)
        want = torch.from_numpy(np.diff(rp)[plan.order_rid].astype(np.float64)).cuda()
        got = y.double()
        fine = (got - want).abs() <= (0.0 if prec == 64 else 1e-2) * want.clamp(min=1)
        ok = bool(torch.where(want > 65504.0, torch.isinf(got) | fine, fine).all().item()) if prec == 16 else bool(fine.all().item())
        st = plan.stats
        b_alg = ci.size * (prec // 8 + 4) + (m + 1) * 4 + (n + m) * (prec // 8)
        form = "two-phase" if st["two_phase"] else ("%d panels" % st["n_col_panels"] if st["n_col_panels"] else ("LDS windows" if st["x_window_on"] else "plain"))
        rows_out.append((desc, prec, m, ci.size, b_alg / 1e6, best * 1e3, b_alg / (best * 1e6) / 8000, st["row_long"], st["row_block"], m - st["row_long"] - st["row_block"], form, ok))
        print("%-70s f%d rows %9d nnz %10d %8.1f MB %9.1f us  %.3f  long/medium/short rows %d/%d/%d %s %s (%.0f s)" % (rows_out[-1][:11] + ("exact" if ok else "WRONG", time.time() - t0)), flush=True)
        plan.close(); del x, y, plan
        torch.cuda.empty_cache()
if len(sys.argv) > 1:
    with open(sys.argv[1], "w") as f:
        f.write("| matrix | dtype | rows | nonzeros | CSR bytes MB | us per SpMV | fraction of 8 TB/s | long / medium / short (+ empty) rows | form | check |\n|---|---|---|---|---|---|---|---|---|---|\n")
        for r in rows_out:
            f.write("| %s | f%d | %d | %d | %.1f | %.1f | %s | %d / %d / %d | %s | %s |\n" % (r[0], r[1], r[2], r[3], r[4], r[5], ("**%.3f**" if r[6] >= 0.6 else "%.3f") % r[6], r[7], r[8], r[9], r[10], "exact" if r[11] else "WRONG"))

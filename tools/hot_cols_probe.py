#!/usr/bin/env python3
"""tools/hot_cols_probe.py -- would clustering the most frequently referenced columns help the gather-bound kernels?  The columns of every range of `span` columns
(a column panel's width, or all of x) are renumbered so that the range's K most frequent columns come first (32 KB of x: what a CU's L1 holds); x is permuted
alike on the host; the plan is built from the renumbered CSR with unchanged options.  Times the SpMV only (the permutation of x would be one more streaming pass)."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import dasp_amd as D
for spec in (sys.argv[1:] or ["rmat_2M:16", "ljournal-2008:16", "webbase-1M:16", "powerlaw_1M:64", "ljournal-2008-uniform:16"]):
    name, prec = spec.split(":"); prec = int(prec)
    rows, cols = D.synth_dims(name, 1.0)
    rp, ci = D.synth_csr(name, 1.0)
    dt, tdt = (np.float64, torch.float64) if prec == 64 else (np.float16, torch.float16)
    val = np.ones(ci.size, dt)
    base = D.Plan(rp, ci, val, cols, precision=prec); P = max(1, base.stats["n_col_panels"])
    line = "%-22s f%d (%d panels):" % (name, prec, P)
    x = torch.ones(cols, dtype=tdt, device="cuda"); y = torch.zeros(rows, dtype=tdt, device="cuda")
    base.upload(); base.drop_host()
    line += " as is %.4f ms" % base.time(x.data_ptr(), y.data_ptr(), 0, 20, 200)[1]; base.close()
    rowid = np.repeat(np.arange(rows, dtype=np.int64), np.diff(rp))
    srt = bool((np.diff(ci.astype(np.int64)) >= 0)[np.diff(rowid) == 0].all())
    o = np.lexsort((ci, rowid))
    p = D.Plan(rp, ci[o], val, cols, precision=prec); p.upload(); p.drop_host()
    line += " | rows' columns ascending already: %s; sorted: %.4f ms" % (srt, p.time(x.data_ptr(), y.data_ptr(), 0, 20, 200)[1]); p.close()
    cnt = np.bincount(ci, minlength=cols)
    for K in (16384 * 2 // (prec // 8) // 2, 4 * 16384 * 2 // (prec // 8) // 2):      # values in 32 KB / 128 KB of x
        bnd = [((cols * k // P + 63) // 64) * 64 for k in range(P)] + [cols]; bnd[0] = 0
        newid = np.empty(cols, np.int64)
        for k in range(P):
            lo, hi = bnd[k], bnd[k + 1]
            c = cnt[lo:hi]
            kk = min(K, hi - lo)
            hot = np.argpartition(-c, kk - 1)[:kk]; hot.sort()
            mask = np.zeros(hi - lo, bool); mask[hot] = True
            order = np.concatenate([np.nonzero(mask)[0], np.nonzero(~mask)[0]])      # old local id at new local position
            newid[lo + order] = lo + np.arange(hi - lo)
        ci2 = newid[ci].astype(np.int32)
        # keep every row's columns ascending, as a CSR from a file would be
        o = np.lexsort((ci2, rowid)); ci2 = ci2[o]
        p = D.Plan(rp, ci2, val, cols, precision=prec, col_panels=P if P > 1 else 1); p.upload(); p.drop_host()
        cov = cnt[np.argsort(-cnt)[:K * P]].sum() / ci.size
        line += " | hottest %d per range first (<= %.0f %% of the gathers): %.4f ms" % (K, 100 * cov, p.time(x.data_ptr(), y.data_ptr(), 0, 20, 200)[1]); p.close()
    print(line, flush=True)

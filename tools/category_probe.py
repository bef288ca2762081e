#!/usr/bin/env python3
"""tools/category_probe.py <workload> <prec> -- the rows of one length class alone (all other rows emptied), ms per SpMV: which class carries a small matrix's time."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import dasp_amd as D
name, prec = sys.argv[1], int(sys.argv[2])
rows, cols = D.synth_dims(name, 1.0)
rp, ci = D.synth_csr(name, 1.0)
lens = np.diff(rp)
dt, tdt = (np.float64, torch.float64) if prec == 64 else (np.float16, torch.float16)
x = torch.ones(cols, dtype=tdt, device="cuda"); y = torch.zeros(rows, dtype=tdt, device="cuda")
rowid = np.repeat(np.arange(rows), lens)
def run(label, keep_rows):
    keep = keep_rows[rowid]
    l2 = np.where(keep_rows, lens, 0)
    rp2 = np.zeros(rows + 1, np.int64); np.cumsum(l2, out=rp2[1:])
    p = D.Plan(rp2.astype(np.int32), ci[keep], np.ones(int(keep.sum()), dt), cols, precision=prec); st = p.stats
    p.upload()
    t = p.time(x.data_ptr(), y.data_ptr(), 0, 20, 300)[1]
    print("%-28s rows %8d nnz %9d: %.4f ms   (blocks %d, pieces %d, short tiles %d, as pieces %d, windows %d)" % (label, int(keep_rows.sum()), int(keep.sum()), t, st["n_med_blocks"], st["n_long_pieces"], st["n_short_tiles"], st["med_rows_as_pieces"], st["n_windows"]), flush=True)
    p.close()
run("all", np.ones(rows, bool))
run("no rows at all", np.zeros(rows, bool))
for lo, hi in ((1, 4), (5, 8), (9, 16), (17, 32), (33, 64), (65, 255), (256, 10**9), (5, 255), (1, 255), (5, 10**9)):
    run("lengths %d..%d" % (lo, hi), (lens >= lo) & (lens <= hi))

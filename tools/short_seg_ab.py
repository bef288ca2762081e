#!/usr/bin/env python3
"""tools/short_seg_ab.py: short rows as slabs (short_seg = -1) against the wave-segmented DPP layout (short_seg = 1), same process, interleaved rounds."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import dasp_amd as D
for name, prec, scale in (("webbase-1M", 64, 1.0), ("webbase-1M", 16, 1.0), ("powerlaw_1M", 64, 1.0), ("ljournal-2008", 16, 1.0), ("rmat_2M", 16, 1.0), ("webbase-1M-uniform", 64, 1.0), ("nlpkkt160", 64, 1.0)):
    rows, cols = D.synth_dims(name, scale)
    rp, ci = D.synth_csr(name, scale)
    dt, tdt = (np.float64, torch.float64) if prec == 64 else (np.float16, torch.float16)
    v = np.ones(ci.size, dt)
    plans = {}
    for seg in (-1, 1):
        p = D.Plan(rp, ci, v, cols, precision=prec, short_seg=seg).upload(); p.drop_host(); plans[seg] = p
    x = torch.ones(cols, dtype=tdt, device="cuda"); y = torch.zeros(rows, dtype=tdt, device="cuda")
    res = {-1: [], 1: []}
    for rnd in range(3):
        for seg in (-1, 1):
            res[seg].append(plans[seg].time(x.data_ptr(), y.data_ptr(), 0, 20, 200)[1])
    st = plans[1].stats
    print("%-20s f%d short rows %d of %d (nnz_short %d of %d): slabs %s ms | wave-segmented %s ms" % (name, prec, rows - st["row_long"] - st["row_block"] - st["row_zero"], rows, st["nnz_short"], st["nnzA"],
          " ".join("%.4f" % t for t in res[-1]), " ".join("%.4f" % t for t in res[1])), flush=True)
    for p in plans.values(): p.close()
    del x, y

#!/usr/bin/env python3
"""tools/panel_store_bound.py (experiment build: DASP_AMD_SO=dasp_amd/variants/exp/libdasp_amd.so): what could ANY better way of writing the column
panels' partial results gain?  The same plans with every y store compiled out of the kernels (DASP_YSTORE=3; results are garbage): the bound
VERDICT r3 next #3 (one shared row order for all panels) is after."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import dasp_amd as D
for name, prec in (("ljournal-2008", 16), ("ljournal-2008-uniform", 16), ("powerlaw_1M", 64), ("webbase-1M", 16), ("rmat_2M", 16)):
    rows, cols = D.synth_dims(name, 1.0)
    rp, ci = D.synth_csr(name, 1.0)
    dt, tdt = (np.float64, torch.float64) if prec == 64 else (np.float16, torch.float16)
    p = D.Plan(rp, ci, np.ones(ci.size, dt), cols, precision=prec).upload(); p.drop_host()
    x = torch.ones(cols, dtype=tdt, device="cuda"); y = torch.zeros(rows, dtype=tdt, device="cuda")
    out = []
    for mode in (0, 3, 4, 0):
        os.environ["DASP_YSTORE"] = str(mode)
        out.append(p.time(x.data_ptr(), y.data_ptr(), 0, 20, 200)[1])
    os.environ["DASP_YSTORE"] = "0"
    print("%-22s f%d panels %d: plain %.4f ms | no y / partial-y store at all %.4f | every store into one 4-KiB window %.4f | plain again %.4f" % (name, prec, p.stats["n_col_panels"], *out), flush=True)
    p.close(); del x, y

#!/usr/bin/env python3
"""tools/panel_count_ab.py -- the number of column panels with row tiles on (the r2 sweep that set ~2.75 MiB of x per panel predates them)."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import dasp_amd as D
counts = [int(b) for b in os.environ.get("PANEL_COUNTS", "0,2,3,4,5,6,8").split(",")]
for spec in (sys.argv[1:] or ["ljournal-2008:16", "ljournal-2008-uniform:16", "powerlaw_1M:64"]):
    name, prec = spec.split(":"); prec = int(prec)
    rows, cols = D.synth_dims(name, 1.0)
    rp, ci = D.synth_csr(name, 1.0)
    dt, tdt = (np.float64, torch.float64) if prec == 64 else (np.float16, torch.float16)
    val = np.ones(ci.size, dt)
    x = torch.ones(cols, dtype=tdt, device="cuda"); y = torch.zeros(rows, dtype=tdt, device="cuda")
    line = "%-22s f%d:" % (name, prec)
    for P in counts:
        for T in (0, -1):
            p = D.Plan(rp, ci, val, cols, precision=prec, col_panels=P, row_tile_max=T); st = p.stats
            p.upload(); p.drop_host()
            line += "  P=%d%s %.4f" % (st["n_col_panels"], "" if T == 0 else " (no tiles)", p.time(x.data_ptr(), y.data_ptr(), 0, 20, 200)[1])
            p.close()
    print(line, flush=True)

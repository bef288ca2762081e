#!/usr/bin/env python3
"""tools/panel_order_probe.py -- column panels with their row tiles in the parent's slot order (y permuted) against row order (y natural): does the
order in which the tiles walk the rows change the gathers' L1 hits?"""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import dasp_amd as D
for spec in (sys.argv[1:] or ["ljournal-2008:16", "ljournal-2008-uniform:16", "powerlaw_1M:64"]):
    name, prec = spec.split(":"); prec = int(prec)
    rows, cols = D.synth_dims(name, 1.0)
    rp, ci = D.synth_csr(name, 1.0)
    dt, tdt = (np.float64, torch.float64) if prec == 64 else (np.float16, torch.float16)
    val = np.ones(ci.size, dt)
    x = torch.ones(cols, dtype=tdt, device="cuda"); y = torch.zeros(rows, dtype=tdt, device="cuda")
    line = "%-22s f%d:" % (name, prec)
    for yo in (0, 1):
        for T in (0, -1, 64):
            p = D.Plan(rp, ci, val, cols, precision=prec, y_order=yo, row_tile_max=min(T, 32)); st = p.stats
            p.upload(); p.drop_host()
            line += "  %s T=%d: %.4f" % ("permuted" if yo == 0 else "natural", st["row_tile_max"], p.time(x.data_ptr(), y.data_ptr(), 0, 20, 200)[1])
            p.close()
    print(line, flush=True)

#!/usr/bin/env python3
"""tools/row_tile_ab.py -- the column panels' row tiles (dasp_options_t::row_tile_max) against the panels without them, same device, same process:
ms per SpMV for every bound asked for (-1 = off)."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import dasp_amd as D
bounds = [int(b) for b in os.environ.get("ROW_TILE_BOUNDS", "-1,6,8,12,16,-1").split(",")]
for spec in (sys.argv[1:] or ["ljournal-2008:16", "ljournal-2008-uniform:16", "powerlaw_1M:64", "powerlaw_1M:16", "ljournal-2008:64"]):
    name, prec = spec.split(":"); prec = int(prec)
    rows, cols = D.synth_dims(name, 1.0)
    rp, ci = D.synth_csr(name, 1.0)
    dt, tdt = (np.float64, torch.float64) if prec == 64 else (np.float16, torch.float16)
    val = np.random.default_rng(3).uniform(0.5, 1.5, ci.size).astype(dt)
    x = torch.from_numpy(np.random.default_rng(4).uniform(0.5, 1.5, cols).astype(dt)).cuda()
    y = torch.zeros(rows, dtype=tdt, device="cuda")
    line, first = "%-22s f%d:" % (name, prec), None
    for b in bounds:
        p = D.Plan(rp, ci, val, cols, precision=prec, row_tile_max=b)
        st = p.stats
        p.upload(); p.drop_host()
        t = p.time(x.data_ptr(), y.data_ptr(), 0, 20, 200)[1]
        got = y.float().cpu().numpy()
        if first is None: first = got
        line += "  T=%d %.4f ms (%d panels, %.0f %% of nnz in tiles, max |dy|/|y| %.1e)" % (b, t, st["n_col_panels"], 100.0 * st["row_tile_nnz"] / max(1, ci.size), float(np.max(np.abs(got - first) / np.maximum(1, np.abs(first)))))
        p.close()
    print(line, flush=True)

#!/usr/bin/env python3
"""tools/one_piece_ab.py -- long rows of up to 4096 nonzeros as ONE piece (no dasp_long_reduce launch) against pieces of 1024: DASP_ONE_PIECE_MAX=0 / default."""
# NOTE (r5): the packers read their A/B environment knobs once per process now -- run one process per setting.
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import dasp_amd as D
for spec in (sys.argv[1:] or ["ljournal-2008:16", "ljournal-2008-uniform:16", "ljournal-2008:64", "webbase-1M:16", "webbase-1M:64"]):
    name, prec = spec.split(":"); prec = int(prec)
    rows, cols = D.synth_dims(name, 1.0)
    rp, ci = D.synth_csr(name, 1.0)
    dt, tdt = (np.float64, torch.float64) if prec == 64 else (np.float16, torch.float16)
    val = np.ones(ci.size, dt)
    x = torch.ones(cols, dtype=tdt, device="cuda"); y = torch.zeros(rows, dtype=tdt, device="cuda")
    line = "%-22s f%d:" % (name, prec)
    for knob in ("0", "4096", "0", "4096"):
        os.environ["DASP_ONE_PIECE_MAX"] = knob
        p = D.Plan(rp, ci, val, cols, precision=prec); st = p.stats
        p.upload(); p.drop_host()
        line += "  one piece <= %s: %.4f ms (%d rows in several pieces)" % (knob, p.time(x.data_ptr(), y.data_ptr(), 0, 20, 200)[1], st["n_long_multi"])
        p.close()
    print(line, flush=True)

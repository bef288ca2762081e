#!/usr/bin/env python3
"""tools/launch_floor_probe.py -- what one SpMV launch costs before any nonzero is read: plans of n empty rows / n rows of one nonzero, back to back like the timing protocol."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import dasp_amd as D
for prec in (64, 16):
    dt, tdt = (np.float64, torch.float64) if prec == 64 else (np.float16, torch.float16)
    line = "f%d:" % prec
    for m, L in ((64, 0), (64, 1), (121192, 0), (121192, 1), (1000005, 0), (1000005, 1)):
        rp = (np.arange(m + 1, dtype=np.int64) * L).astype(np.int32)
        ci = (np.arange(m * L, dtype=np.int64) % m).astype(np.int32)
        p = D.Plan(rp, ci, np.ones(ci.size, dt), m, precision=prec).upload()
        x = torch.ones(m, dtype=tdt, device="cuda"); y = torch.zeros(m, dtype=tdt, device="cuda")
        line += "  %d rows x %d: %.2f us" % (m, L, 1e3 * p.time(x.data_ptr(), y.data_ptr(), 0, 100, 1000)[1])
        p.close()
    print(line, flush=True)

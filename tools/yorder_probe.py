#!/usr/bin/env python3
"""tools/yorder_probe.py <workload> <precision> [scale]: the same plan with y in the reference's permuted order and in natural order.  For column-panel
plans the difference is where the panels' partial results are written: the parent's slot order (scattered 2- / 8-byte stores, every panel sorts its
rows differently) or row order (a slab's rows are increasing row ids: neighbouring stores share sectors)."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import dasp_amd as D
name, prec = sys.argv[1], int(sys.argv[2])
scale = float(sys.argv[3]) if len(sys.argv) > 3 else 1.0
m, n = D.synth_dims(name, scale)
rp, ci = D.synth_csr(name, scale)
dt, tdt = (np.float64, torch.float64) if prec == 64 else (np.float16, torch.float16)
v = np.ones(ci.size, dt)
x = torch.ones(n, dtype=tdt, device="cuda"); y = torch.zeros(m, dtype=tdt, device="cuda")
for tag, kw in (("permuted", dict(y_order=0)), ("natural", dict(y_order=1)), ("permuted, no panels", dict(y_order=0, col_panels=1)), ("natural, no panels", dict(y_order=1, col_panels=1))):
    p = D.Plan(rp, ci, v, n, precision=prec, **kw).upload()
    w, e = p.time(x.data_ptr(), y.data_ptr(), 0, warmup=10, iters=100)
    print("%s f%d %-22s panels=%d  %.4f ms" % (name, prec, tag, p.stats["n_col_panels"], e), flush=True)
    p.close()

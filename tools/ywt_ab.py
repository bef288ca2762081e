#!/usr/bin/env python3
"""tools/ywt_ab.py: y stores written through (the default for f64 plans that stream from HBM) against plain stores (DASP_Y_WT=0), same plan, same vectors, interleaved."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import dasp_amd as D
for name in ("HV15R", "Queen_4147", "nlpkkt160", "HV15R-unstructured"):
    rows, cols = D.synth_dims(name, 1.0)
    rp, ci = D.synth_csr(name, 1.0)
    plans = [D.Plan(rp, ci, np.ones(ci.size), cols).upload() for _ in range(2)]
    for p in plans: p.drop_host()
    x = torch.ones(cols, dtype=torch.float64, device="cuda")
    ys = [torch.zeros(rows, dtype=torch.float64, device="cuda") for _ in range(2)]
    for k, p in enumerate(plans):
        for j, y in enumerate(ys):
            line = "%-20s plan %d y %d:" % (name, k, j)
            for rnd in range(2):
                for wt in ("1", "0"):
                    os.environ["DASP_Y_WT"] = wt
                    line += "  %s %.4f" % ("written through" if wt == "1" else "plain", p.time(x.data_ptr(), y.data_ptr(), 0, 10, 200)[1])
            print(line, flush=True)
    os.environ.pop("DASP_Y_WT")
    for p in plans: p.close()
    del x, ys

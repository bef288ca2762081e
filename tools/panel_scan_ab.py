#!/usr/bin/env python3
"""tools/panel_scan_ab.py: column panels whose classifier walks the rows in the parent's slot order (r4) against row order (r3, DASP_PANEL_ROW_SCAN=1)."""
# NOTE (r5): the packers read their A/B environment knobs once per process now -- run one process per setting.
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import dasp_amd as D
for name, prec in (("ljournal-2008", 16), ("ljournal-2008-uniform", 16), ("powerlaw_1M", 64)):
    rows, cols = D.synth_dims(name, 1.0)
    rp, ci = D.synth_csr(name, 1.0)
    dt, tdt = (np.float64, torch.float64) if prec == 64 else (np.float16, torch.float16)
    plans = {}
    for tag, env in (("row order", "1"), ("parent slot order", None)):
        if env: os.environ["DASP_PANEL_ROW_SCAN"] = env
        else: os.environ.pop("DASP_PANEL_ROW_SCAN", None)
        p = D.Plan(rp, ci, np.ones(ci.size, dt), cols, precision=prec).upload(); p.drop_host(); plans[tag] = p
    x = torch.ones(cols, dtype=tdt, device="cuda"); y = torch.zeros(rows, dtype=tdt, device="cuda")
    res = {k: [] for k in plans}
    for rnd in range(3):
        for k, p in plans.items(): res[k].append(p.time(x.data_ptr(), y.data_ptr(), 0, 20, 200)[1])
    print("%-22s f%d panels %d: " % (name, prec, plans["row order"].stats["n_col_panels"]) + " | ".join("%s %s ms" % (k, " ".join("%.4f" % t for t in v)) for k, v in res.items()), flush=True)
    for p in plans.values(): p.close()
    del x, y

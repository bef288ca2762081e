#!/usr/bin/env python3
"""tools/y_in_arena_probe.py (experiment build): the written vector INSIDE the plan's own allocation -- is a y that shares the arena's allocation always in the
fast class (or always in the slow one)?  Three uploads of one plan, two outside y vectors each, against y (and x) placed behind the arena's arrays."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import dasp_amd as D
for name in (sys.argv[1:] or ["HV15R", "nlpkkt160", "Queen_4147"]):
    rows, cols = D.synth_dims(name, 1.0)
    rp, ci = D.synth_csr(name, 1.0)
    plans = [D.Plan(rp, ci, np.ones(ci.size), cols).upload() for _ in range(3)]
    for p in plans: p.drop_host()
    del ci
    x = torch.ones(cols, dtype=torch.float64, device="cuda")
    ys = [torch.zeros(rows, dtype=torch.float64, device="cuda") for _ in range(2)]
    for k, p in enumerate(plans):
        line = "%-12s plan %d:" % (name, k)
        for j, y in enumerate(ys):
            os.environ["DASP_Y_IN_ARENA"] = "0"
            line += "  outside y %d %.4f" % (j, p.time(x.data_ptr(), y.data_ptr(), 0, 10, 200)[1])
        for mode, label in ((1, "y in the arena"), (2, "x and y in the arena"), (1, "y in the arena again")):
            os.environ["DASP_Y_IN_ARENA"] = str(mode)
            line += "  | %s %.4f" % (label, p.time(x.data_ptr(), ys[0].data_ptr(), 0, 10, 200)[1])
        os.environ["DASP_Y_IN_ARENA"] = "0"
        print(line, flush=True)
    for p in plans: p.close()
    del x, ys

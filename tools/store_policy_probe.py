#!/usr/bin/env python3
"""tools/store_policy_probe.py (experiment build): every cache-policy combination of the y store (sc0 / sc1 / nt bits written in assembly) in a kernel whose table
reads do not depend on the compiler's alias analysis (dasp_spmv_kt_kernel).  First line: that kernel's plain mode against the product kernel of the same build."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import dasp_amd as D
MODES = ((0, "plain"), (10, "sc1"), (11, "sc0"), (12, "sc0 sc1"), (13, "nt"), (14, "sc1 nt"), (15, "sc0 sc1 nt"), (16, "sc0 nt"), (3, "no store"), (0, "plain again"))
for name in (sys.argv[1:] or ["HV15R", "nlpkkt160", "Queen_4147"]):
    rows, cols = D.synth_dims(name, 1.0)
    rp, ci = D.synth_csr(name, 1.0)
    plans = [D.Plan(rp, ci, np.ones(ci.size), cols, cid8=-1 if name == "nlpkkt160" else 0).upload() for _ in range(2)]
    for p in plans: p.drop_host()
    del ci
    x = torch.ones(cols, dtype=torch.float64, device="cuda")
    ys = [torch.zeros(rows, dtype=torch.float64, device="cuda") for _ in range(2)]
    for k, p in enumerate(plans):
        for j, y in enumerate(ys):
            os.environ["DASP_KT_KERNEL"] = "0"; os.environ["DASP_YSTORE"] = "0"; os.environ["DASP_Y_WT"] = "0"
            line = "%-11s plan %d y %d: product kernel %.4f |" % (name, k, j, p.time(x.data_ptr(), y.data_ptr(), 0, 10, 100)[1])
            os.environ["DASP_KT_KERNEL"] = "1"
            for mode, label in MODES:
                os.environ["DASP_YSTORE"] = str(mode)
                line += " %s %.4f" % (label, p.time(x.data_ptr(), y.data_ptr(), 0, 10, 100)[1])
            print(line, flush=True)
    for p in plans: p.close()
    del x, ys

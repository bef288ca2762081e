#!/usr/bin/env python3
"""tools/cid8_ab.py: one-byte ids on / off on the FEM stand-ins (r3: the 16-bit-id kernel without them runs at 72 registers / 7 waves per SIMD, the one-byte kernel at 78 / 6)"""
import sys, os
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import dasp_amd as D
for name in sys.argv[1:] or ["HV15R", "Queen_4147", "HV15R-unstructured"]:
    rows, cols = D.synth_dims(name, 1.0)
    rp, ci = D.synth_csr(name, 1.0)
    v = np.ones(ci.size)
    x = torch.ones(cols, dtype=torch.float64, device="cuda"); y = torch.zeros(rows, dtype=torch.float64, device="cuda")
    plans = {}
    order = (("cid8 on", dict()), ("cid8 off", dict(cid8=-1)))
    if os.environ.get("AB_SWAP"): order = order[::-1]
    for tag, kw in order:
        p = D.Plan(rp, ci, v, cols, **kw).upload(); p.drop_host(); plans[tag] = p
    for rnd in range(3):
        for tag, p in plans.items():
            e = p.time(x.data_ptr(), y.data_ptr(), 0, 20, 200)[1]
            print("%-20s %-9s round %d: %.4f ms  (cid8 chunks %d)" % (name, tag, rnd, e, p.stats["cid8_chunks"]), flush=True)
    for p in plans.values(): p.close()
    del x, y
    torch.cuda.empty_cache()

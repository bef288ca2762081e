#!/usr/bin/env python3
"""tools/cid8_oneshot_ab.py: one-byte ids in one-shot f64 blocks (r4) against 16-bit ids there (cid8 = -1 turns all one-byte ids off: on nlpkkt160 every
block is one-shot, so that is the r3 layout), same process, interleaved."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import dasp_amd as D
for name in ("nlpkkt160",):
    rows, cols = D.synth_dims(name, 1.0)
    rp, ci = D.synth_csr(name, 1.0)
    v = np.ones(ci.size)
    plans = {}
    for tag, c8 in (("16-bit ids", -1), ("one-byte ids", 0)):
        p = D.Plan(rp, ci, v, cols, cid8=c8).upload(); p.drop_host(); plans[tag] = p
        st = p.stats
        print("%s %s: cid8 chunks %d of %d, data_X %.1f MB" % (name, tag, st["cid8_chunks"], st["fill0_nnz_reg"] // 64, st["data_X"] / 1e6), flush=True)
    x = torch.ones(cols, dtype=torch.float64, device="cuda")
    ys = [torch.zeros(rows, dtype=torch.float64, device="cuda") for _ in range(2)]
    for j, y in enumerate(ys):
        res = {k: [] for k in plans}
        for rnd in range(3):
            for k, p in plans.items(): res[k].append(p.time(x.data_ptr(), y.data_ptr(), 0, 10, 200)[1])
        print("%s y %d: " % (name, j) + " | ".join("%s %s ms" % (k, " ".join("%.4f" % t for t in vv)) for k, vv in res.items()), flush=True)

#!/usr/bin/env python3
"""tools/placement_probe.py [workload=HV15R] [copies=6]: the SAME plan uploaded several times (one arena allocation each, all alive together), timed interleaved:
do two uploads of identical bytes run at different speeds (physical placement of the arena in HBM)?"""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import dasp_amd as D
name = sys.argv[1] if len(sys.argv) > 1 else "HV15R"
copies = int(sys.argv[2]) if len(sys.argv) > 2 else 6
prec = int(sys.argv[3]) if len(sys.argv) > 3 else 64
rows, cols = D.synth_dims(name, 1.0)
rp, ci = D.synth_csr(name, 1.0)
if os.environ.get("PROBE_ONE_COLUMN"):      # every gather reads x[0]: the stream of the arena alone
    ci = np.zeros_like(ci)
v = np.ones(ci.size, np.float64 if prec == 64 else np.float16)
dt = torch.float64 if prec == 64 else torch.float16
plans = []
for k in range(copies):
    p = D.Plan(rp, ci, v, cols, precision=prec).upload(); p.drop_host(); plans.append(p)
xs = [torch.ones(cols, dtype=dt, device="cuda") for _ in range(2)]
ys = [torch.zeros(rows + 64, dtype=dt, device="cuda") for _ in range(2)]
print("x buffers %s, y buffers %s" % ([hex(t.data_ptr()) for t in xs], [hex(t.data_ptr()) for t in ys]), flush=True)
IT = int(os.environ.get("PROBE_ITERS", "200"))
for rnd in range(int(os.environ.get("PROBE_ROUNDS", "4"))):
    line = "%s round %d:" % (name, rnd)
    for k, p in enumerate(plans):
        e = p.time(xs[rnd % 2].data_ptr(), ys[rnd % 2].data_ptr(), 0, 10 if IT >= 100 else 0, IT)[1]
        line += "  plan %d %.4f" % (k, e)
    print(line, flush=True)

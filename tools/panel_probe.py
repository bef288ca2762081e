#!/usr/bin/env python3
"""Column-panel sweep: time a workload with col_panels = 1, 2, 4, ... under both cache policies.
usage: panel_probe.py <workload> <precision> [scale]      (results: DESIGN.md section 4, "column panels")
"""
import sys

import numpy as np
import torch

import dasp_amd as D

name, prec = sys.argv[1], int(sys.argv[2])
scale = float(sys.argv[3]) if len(sys.argv) > 3 else 1.0
m, n = D.synth_dims(name, scale)
rp, ci = D.synth_csr(name, scale)
val = np.ones(ci.size, np.float64 if prec == 64 else np.float16)
x = torch.ones(n, dtype=torch.float64 if prec == 64 else torch.float16, device="cuda")
y = torch.zeros(m, dtype=x.dtype, device="cuda")
for P in (1, 2, 3, 4, 6, 8, 12, 16):
    plan = D.Plan(rp, ci, val, n, precision=prec, col_panels=P).upload()
    plan.drop_host()
    out = []
    for pol in (1, 2):
        plan.set_stream_policy(pol)
        _, ev = plan.time(x.data_ptr(), y.data_ptr(), 0, 20, 100)
        out.append(ev)
    ok = bool((y.double().cpu().numpy() == np.diff(rp)[plan.order_rid]).all()) if prec == 64 else None
    print(f"{name} f{prec} col_panels={P}: plain {out[0]:.4f} ms, non-temporal {out[1]:.4f} ms  pre {plan.stats['pre_ms']:.0f} ms exact={ok}", flush=True)
    plan.close()

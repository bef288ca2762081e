#!/usr/bin/env python3
"""tools/devpre_laps.py <workload> <precision>: dasp_plan_create_device with DASP_VERBOSE=1 (lap times of every stage, every panel) + wall time of two builds"""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["DASP_VERBOSE"] = "1"
import dasp_amd as D
name, prec = sys.argv[1], int(sys.argv[2])
rows, cols = D.synth_dims(name, 1.0)
rp, ci = D.synth_csr(name, 1.0)
nnz = int(rp[-1])
d_rp = torch.from_numpy(rp).cuda(); d_ci = torch.from_numpy(ci).cuda()
d_v = torch.ones(nnz, dtype=torch.float64 if prec == 64 else torch.float16, device="cuda")
for k in range(2):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    p = D.Plan.from_device(d_rp.data_ptr(), d_ci.data_ptr(), d_v.data_ptr(), rows, cols, nnz, precision=prec)
    torch.cuda.synchronize(); t1 = time.perf_counter()
    print("== build %d: %.1f ms, panels %d" % (k, (t1 - t0) * 1e3, p.stats.get("n_panels", -1)), flush=True)
    p.close()

#!/usr/bin/env python3
"""Fraction of the HBM roofline for 8 M rows of ONE length, as MFMA blocks (slab_max_len=4) and as uniform-length slabs (slab_max_len=32):
where the crossover between the two layouts lies.   usage: rowlen_probe.py [precision ...]"""
import sys

import numpy as np
import torch

import dasp_amd as D

m = 8_000_000
precs = [int(a) for a in sys.argv[1:]] or [64, 16]
for prec in precs:
    dt = np.float64 if prec == 64 else np.float16
    vb = prec // 8
    for L in (5, 6, 8, 10, 12, 16, 20, 24, 32):
        rp = (np.arange(m + 1, dtype=np.int64) * L).astype(np.int32)
        base = [0, 1, -1, 2000, -2000, 2, -2, 4000, -4000, 3, -3, 6000, -6000, 4, -4, 8000]
        offs = np.array((base + [b + 20000 for b in base])[:L], np.int64)      # stencil-like: near and far neighbours
        ci = (np.arange(m, dtype=np.int64)[:, None] + offs[None, :]) % m
        ci.sort(axis=1)
        ci = ci.reshape(-1).astype(np.int32)
        val = np.ones(ci.size, dt)
        x = torch.ones(m, dtype=torch.float64 if prec == 64 else torch.float16, device="cuda")
        y = torch.zeros(m, dtype=x.dtype, device="cuda")
        balg = ci.size * (vb + 4) + (m + 1) * 4 + 2 * m * vb
        out = []
        for smax in (4, 32):
            plan = D.Plan(rp, ci, val, m, precision=prec, x_window=-1, slab_max_len=smax).upload()
            plan.drop_host()
            _, e = plan.time(x.data_ptr(), y.data_ptr(), 0, 10, 60)
            ok = bool((y.double().cpu().numpy() == L).all())
            out.append(f"{'blocks' if smax == 4 else 'slabs'} {e*1e3:7.1f} us = {balg/(e*1e-3)/8e12:.3f}{'' if ok else ' WRONG'}")
            plan.close()
        print(f"f{prec} len {L:2d}: " + "   ".join(out), flush=True)
        del ci, val

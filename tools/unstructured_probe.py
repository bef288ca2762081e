#!/usr/bin/env python3
"""Unstructured-FEM-like matrix: every node couples with `deg` random nodes inside a band (what RCM leaves of a tetrahedral mesh), `dof`
unknowns per node, so a row is `deg` runs of `dof` adjacent columns at scattered positions.   usage: unstructured_probe.py [precision=64]"""
import sys

import numpy as np
import torch

import dasp_amd as D

prec = int(sys.argv[1]) if len(sys.argv) > 1 else 64
dt = np.float64 if prec == 64 else np.float16
vb = prec // 8
rng = np.random.default_rng(2)
for nodes, deg, dof, band in ((700_000, 27, 3, 20_000), (700_000, 27, 3, 2_000), (400_000, 15, 5, 50_000), (2_000_000, 14, 1, 30_000)):
    nb = rng.integers(-band, band + 1, size=(nodes, deg)) + np.arange(nodes)[:, None]
    nb = np.clip(nb, 0, nodes - 1)
    nb[:, 0] = np.arange(nodes)
    nb.sort(axis=1)
    cols = (nb[:, :, None] * dof + np.arange(dof)[None, None, :]).reshape(nodes, deg * dof)      # node block -> dof columns
    ci = np.repeat(cols, dof, axis=0).reshape(-1).astype(np.int32)                                 # every dof row of the node
    m = nodes * dof
    L = deg * dof
    rp = (np.arange(m + 1, dtype=np.int64) * L).astype(np.int32)
    val = np.ones(ci.size, dt)
    x = torch.ones(m, dtype=torch.float64 if prec == 64 else torch.float16, device="cuda")
    y = torch.zeros(m, dtype=x.dtype, device="cuda")
    balg = ci.size * (vb + 4) + (m + 1) * 4 + 2 * m * vb
    for kw in (dict(), dict(x_window=-1, cid16=-1)):
        plan = D.Plan(rp, ci, val, m, precision=prec, **kw).upload()
        plan.drop_host()
        _, e = plan.time(x.data_ptr(), y.data_ptr(), 0, 10, 100)
        st = plan.stats
        print(f"nodes {nodes} deg {deg} dof {dof} band +-{band} f{prec} {kw}: rows {m} nnz {ci.size} {e*1e3:8.1f} us = {balg/(e*1e-3)/8e12:.3f} of 8 TB/s "
              f"(windows {st['n_windows_lds']}/{st['n_windows']} cid16 {st['cid16_on']} panels {st['n_col_panels']} blocks {st['n_med_blocks']})", flush=True)
        plan.close()
    del ci, val, cols, nb

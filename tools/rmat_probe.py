#!/usr/bin/env python3
"""R-MAT graph (a, b, c, d = 0.57, 0.19, 0.19, 0.05: the Graph500 generator, a closer proxy for web / social graphs than uniform
columns) through the automatic plan choices and the column-panel settings.   usage: rmat_probe.py [scale=21] [edge_factor=16] [precision=16]"""
import sys

import numpy as np
import scipy.sparse as sp
import torch

import dasp_amd as D

scale = int(sys.argv[1]) if len(sys.argv) > 1 else 21
ef = int(sys.argv[2]) if len(sys.argv) > 2 else 16
prec = int(sys.argv[3]) if len(sys.argv) > 3 else 16
n = 1 << scale
ne = n * ef
rng = np.random.default_rng(1)
src = np.zeros(ne, np.int64)
dst = np.zeros(ne, np.int64)
for bit in range(scale):
    r = rng.random(ne)
    sb = r >= 0.76                       # c + d: source bit set
    db = ((r >= 0.57) & (r < 0.76)) | (r >= 0.95)     # b or d: destination bit set
    src |= sb.astype(np.int64) << bit
    dst |= db.astype(np.int64) << bit
A = sp.csr_matrix((np.ones(ne, np.float32), (src, dst)), shape=(n, n))
A.sum_duplicates()
A.sort_indices()
rp, ci = A.indptr.astype(np.int32), A.indices.astype(np.int32)
dt = np.float64 if prec == 64 else np.float16
val = np.ones(ci.size, dt)
lens = np.diff(rp)
print(f"R-MAT scale {scale}: {n} rows, {ci.size} nnz, max row {lens.max()}, empty rows {(lens == 0).sum()}", flush=True)
x = torch.ones(n, dtype=torch.float64 if prec == 64 else torch.float16, device="cuda")
y = torch.zeros(n, dtype=x.dtype, device="cuda")
for cp in (0, 1, 2, 3, 4, 6):
    plan = D.Plan(rp, ci, val, n, precision=prec, col_panels=cp).upload()
    plan.drop_host()
    _, e = plan.time(x.data_ptr(), y.data_ptr(), 0, 10, 100)
    st = plan.stats
    ok = bool((y.double().cpu().numpy() == lens[plan.order_rid]).all()) if prec == 64 else None
    print(f"col_panels={cp}: {e*1e3:9.2f} us  panels={st['n_col_panels']} windows={st['n_windows_lds']}/{st['n_windows']} cid16={st['cid16_on']} "
          f"gathers/s={ci.size/e/1e6:.0f} G  exact={ok}", flush=True)
    plan.close()

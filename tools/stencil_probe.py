#!/usr/bin/env python3
"""Short-row PDE stencils (2-D 5-point, 3-D 7-point, 3-D 27-point) through the default plan: achieved fraction of the HBM roofline
and the same matrix through rocSPARSE-like accounting (B_alg = CSR bytes).   usage: stencil_probe.py [precision=64]"""
import sys

import numpy as np
import scipy.sparse as sp
import torch

import dasp_amd as D

prec = int(sys.argv[1]) if len(sys.argv) > 1 else 64
dt = np.float64 if prec == 64 else np.float16
vb = prec // 8


def lap(dims, full):
    eye = [sp.identity(d, format="csr") for d in dims]
    one = [sp.diags([1, 1, 1], [-1, 0, 1], shape=(d, d), format="csr") for d in dims]
    if full:                      # tensor-product stencil (9 / 27 points)
        A = one[0]
        for k in range(1, len(dims)):
            A = sp.kron(A, one[k], format="csr")
        return A
    A = None
    for k in range(len(dims)):    # cross stencil (5 / 7 points)
        T = sp.diags([1, 1], [-1, 1], shape=(dims[k], dims[k]), format="csr")
        parts = [T if j == k else eye[j] for j in range(len(dims))]
        M = parts[0]
        for q in parts[1:]:
            M = sp.kron(M, q, format="csr")
        A = M if A is None else A + M
    return (A + sp.identity(A.shape[0], format="csr")).tocsr()


for tag, dims, full in (("2-D 5-point 3000^2", (3000, 3000), False), ("3-D 7-point 200^3", (200, 200, 200), False),
                        ("3-D 27-point 160^3", (160, 160, 160), True), ("2-D 9-point 3000^2", (3000, 3000), True)):
    A = lap(dims, full).tocsr()
    A.sort_indices()
    m, n = A.shape
    rp, ci = A.indptr.astype(np.int32), A.indices.astype(np.int32)
    val = np.ones(ci.size, dt)
    x = torch.ones(n, dtype=torch.float64 if prec == 64 else torch.float16, device="cuda")
    y = torch.zeros(m, dtype=x.dtype, device="cuda")
    plan = D.Plan(rp, ci, val, n, precision=prec).upload()
    plan.drop_host()
    _, e = plan.time(x.data_ptr(), y.data_ptr(), 0, 20, 200)
    st = plan.stats
    balg = ci.size * (vb + 4) + (m + 1) * 4 + (m + n) * vb
    ok = bool((y.double().cpu().numpy() == np.diff(rp)[plan.order_rid]).all())
    print(f"{tag} f{prec}: rows {m} nnz {ci.size}  {e*1e3:8.1f} us  {balg/e/1e6:7.0f} GB/s = {balg/(e*1e-3)/8e12:.3f} of 8 TB/s  "
          f"fill0={st['rate_fill0']:.3f} short={m-st['row_long']-st['row_block']} med={st['row_block']} cid16={st['cid16_on']} "
          f"windows={st['n_windows_lds']}/{st['n_windows']} exact={ok}", flush=True)
    plan.close()

#!/usr/bin/env python3
"""Where does a latency-bound stand-in spend its time?  Re-times its medium rows with the column ids replaced by trivially
cacheable ones (what is left is everything but the gathers) and with the fill threshold lowered (no irregular tails).
usage: medium_probe.py <workload> <precision>     (results: DESIGN.md section 4.4)"""
import sys

import numpy as np
import torch

import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import dasp_amd as D

name, prec = sys.argv[1], int(sys.argv[2])
m, n = D.synth_dims(name, 1.0)
rp, ci = D.synth_csr(name, 1.0)
lens = np.diff(rp)
dt = np.float64 if prec == 64 else np.float16
x = torch.ones(n, dtype=torch.float64 if prec == 64 else torch.float16, device="cuda")
y = torch.zeros(m, dtype=x.dtype, device="cuda")
keep = (lens >= 5) & (lens < 256)
l2 = np.where(keep, lens, 0)
rp2 = np.zeros(m + 1, np.int32)
np.cumsum(l2, out=rp2[1:])
ci2 = ci[np.repeat(keep, lens)]
rows2 = np.repeat(np.arange(m), l2)


def run(tag, cols, **kw):
    plan = D.Plan(rp2, cols.astype(np.int32), np.ones(cols.size, dt), n, precision=prec, **kw).upload()
    _, e = plan.time(x.data_ptr(), y.data_ptr(), 0, 200, 2000)
    st = plan.stats
    print(f"{name} f{prec} {tag}: {e*1e3:.2f} us  blocks={st['n_med_blocks']} windows={st['n_windows_lds']}/{st['n_windows']} "
          f"reg={st['fill0_nnz_reg']} irreg={st['nnz_irreg']} nnz={cols.size}", flush=True)
    plan.close()


run("medium rows, original columns", ci2)
run("medium rows, original columns, no windows", ci2, x_window=-1)
run("medium rows, all columns = 0, no windows", np.zeros_like(ci2), x_window=-1)
run("medium rows, columns = row id, no windows", np.minimum(rows2, n - 1), x_window=-1)
run("medium rows, original, threshold 0.01, no windows", ci2, threshold=0.01, x_window=-1)
run("medium rows, columns = 0, threshold 0.01, no windows", np.zeros_like(ci2), threshold=0.01, x_window=-1)
run("no rows at all (launch floor)", np.zeros(0, np.int32)) if False else None

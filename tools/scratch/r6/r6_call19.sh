#!/bin/bash
# r6_call19 -- rows of ONE length 12 in a band + 10 % outliers (tests/test_plan_host.py::test_auto_hybrid_windows...): slabs (the r6 dominant-length rule) against the hybrid windows r5 chose
export PYTHONPATH=$PWD
python3 - <<'PY' 2>&1 | grep -v amdgpu.ids
import numpy as np, torch, dasp_amd as D
rng = np.random.default_rng(23)
for m in (120000, 1000000):
    n = m
    rows = np.repeat(np.arange(m), 12)
    ci = np.where(rng.random(rows.size) < 0.9, np.clip(rows + rng.integers(-500, 501, rows.size), 0, n - 1), rng.integers(0, n, rows.size)).astype(np.int32)
    rp = (np.arange(m + 1, dtype=np.int64) * 12).astype(np.int32)
    for prec in (16, 64):
        dt, tdt = (np.float64, torch.float64) if prec == 64 else (np.float16, torch.float16)
        for kw in ({}, {"slab_max_len": 4}, {"slab_max_len": 24}, {"x_window": -1, "slab_max_len": 4}):
            p = D.Plan(rp, ci, np.ones(ci.size, dt), n, precision=prec, **kw).upload()
            x = torch.ones(n, dtype=tdt, device="cuda"); y = torch.zeros(m, dtype=tdt, device="cuda")
            t = min(p.time(x.data_ptr(), y.data_ptr(), 0, 50, 500)[1] for _ in range(3)) * 1e3
            st = p.stats
            print(m, "f%d" % prec, kw, "windows", st["x_window_on"], "hybrid", st["x_window_hybrid"], "blocks", st["n_med_blocks"], "short tiles", st["n_short_tiles"], "%.2f us" % t, flush=True)
            p.close()
PY

import sys, numpy as np, torch
import dasp_amd as D
rng = np.random.default_rng(2)
for prec, nodes, deg, band in ((16, 2_000_000, 14, 30_000), (16, 2_000_000, 14, 15_000), (64, 2_000_000, 14, 8_000), (64, 2_000_000, 14, 4_000)):
    dt = np.float64 if prec == 64 else np.float16
    vb = prec // 8
    nb = np.clip(rng.integers(-band, band + 1, size=(nodes, deg)) + np.arange(nodes)[:, None], 0, nodes - 1)
    nb.sort(axis=1)
    ci = nb.reshape(-1).astype(np.int32)
    rp = (np.arange(nodes + 1, dtype=np.int64) * deg).astype(np.int32)
    val = np.ones(ci.size, dt)
    x = torch.ones(nodes, dtype=torch.float64 if prec == 64 else torch.float16, device="cuda")
    y = torch.zeros(nodes, dtype=x.dtype, device="cuda")
    balg = ci.size * (vb + 4) + (nodes + 1) * 4 + 2 * nodes * vb
    for kw in (dict(), dict(x_window=-1), dict(x_window=163840), dict(x_window=163840, row_window=512), dict(x_window=163840, row_window=256)):
        plan = D.Plan(rp, ci, val, nodes, precision=prec, **kw).upload()
        plan.drop_host()
        _, e = plan.time(x.data_ptr(), y.data_ptr(), 0, 10, 100)
        st = plan.stats
        print(f"f{prec} band +-{band} {kw}: {e*1e3:8.1f} us = {balg/(e*1e-3)/8e12:.3f}  windows {st['n_windows_lds']}/{st['n_windows']} lds {st['lds_bytes']} R {st['row_window']}", flush=True)
        plan.close()

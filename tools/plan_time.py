#!/usr/bin/env python3
"""tools/plan_time.py <workload> <precision> [scale=1] [opt=value ...] -- one plan of a stand-in through the Python mirror: all-ones exact check, event-timed ms, fraction of the
8 TB/s roofline.  Honours DASP_AMD_SO (another build of the library), so two builds can be compared without LD_PRELOAD:  DASP_AMD_SO=dasp_amd/variants/<tag>/libdasp_amd.so python tools/plan_time.py ..."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import dasp_amd as D
name, prec = sys.argv[1], int(sys.argv[2])
rest = sys.argv[3:]
scale = float(rest.pop(0)) if rest and "=" not in rest[0] else 1.0
kw = {k: int(v) for k, v in (a.split("=") for a in rest)}
m, n = D.synth_dims(name, scale)
rp, ci = D.synth_csr(name, scale)
dt = np.float64 if prec == 64 else np.float16
plan = D.Plan(rp, ci, np.ones(ci.size, dt), n, precision=prec, **kw).upload()
plan.drop_host()
tdt = torch.float64 if prec == 64 else torch.float16
x = torch.ones(n, dtype=tdt, device="cuda"); y = torch.zeros(m, dtype=tdt, device="cuda")
for rep in range(2):
    w, e = plan.time(x.data_ptr(), y.data_ptr(), 0, warmup=30, iters=int(os.environ.get("ITERS", "300")))
want = torch.from_numpy((np.diff(rp) if kw.get("y_order") == 1 else np.diff(rp)[plan.order_rid]).astype(np.float64)).cuda()      # DASP_Y_NATURAL: y in row order
got = y.double()
if prec == 64 or int(np.diff(rp).max()) <= 2048:
    ok = bool((got == want).all().item())
else:      # f16 results: rounded beyond 2048, +inf beyond 65504
    fine = (got - want).abs() <= 1e-2 * want.clamp(min=1)
    ok = bool(torch.where(want > 65504.0, torch.isinf(got) | fine, fine).all().item())
st = plan.stats
b_alg = ci.size * (prec // 8 + 4) + (m + 1) * 4 + (n + m) * (prec // 8)
print("%s f%d %s | %.4f ms = %.3f of the roofline | two_phase %d panels %d fill0 %.4f narrow chunks %d, %.3f B/nnz packed | %s | %s" % (name, prec, kw, e, b_alg / (e * 1e6) / 8000, st["two_phase"], st["n_col_panels"], st["rate_fill0"],
      st["cid8_chunks"], st["data_X"] / max(1, ci.size),
      "exact" if ok else "WRONG", os.environ.get("DASP_AMD_SO", "product")))

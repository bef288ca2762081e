"""tools/r5_hybrid_probe.py <workload> <H> -- r5: the rows of fewer than H nonzeros of an f16 stand-in alone, two-phase against the DASP kernels: what a hybrid plan (hub rows column-blocked, the rest two-phase) could gain (profiles/r05_long_cb.md)"""
import os, sys
import numpy as np, torch
sys.path.insert(0, "/root/repo")
import dasp_amd as D
name = sys.argv[1]; H = int(sys.argv[2])
m, n = D.synth_dims(name, 1.0); rp, ci = D.synth_csr(name, 1.0)
lens = np.diff(rp)
keep = lens < H
lens2 = np.where(keep, lens, 0)
rp2 = np.zeros(m + 1, np.int32); np.cumsum(lens2, out=rp2[1:])
mask = np.repeat(keep, lens)
ci2 = np.ascontiguousarray(ci[mask])
print(name, "rows <", H, ": nnz", ci2.size, "of", ci.size, flush=True)
for kw in (dict(two_phase=1), dict(two_phase=-1)):
    plan = D.Plan(rp2, ci2, np.ones(ci2.size, np.float16), n, precision=16, **kw).upload(); plan.drop_host()
    x = torch.ones(n, dtype=torch.float16, device="cuda"); y = torch.zeros(m, dtype=torch.float16, device="cuda")
    for _ in range(2): w, e = plan.time(x.data_ptr(), y.data_ptr(), 0, warmup=20, iters=200)
    print(kw, "%.4f ms" % e, "panels", plan.stats["n_col_panels"], "tp", plan.stats["two_phase"], flush=True)
    plan.close()

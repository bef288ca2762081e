#!/usr/bin/env python3
"""tools/persist_probe.py (experiment build): resident workgroups that keep their medium blocks' results in LDS and write y in one burst
(dasp_spmv_persist_kernel) against one block per wave with direct stores (dasp_spmv_kt_kernel), per (plan, y) allocation pair."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import dasp_amd as D
def env(**kw):
    for k in ("DASP_KT_KERNEL", "DASP_PERSIST", "DASP_PERSIST2", "DASP_PERSIST_GROUPS", "DASP_KT_REPS", "DASP_HYBRID", "DASP_HYBRID_PER_CU", "DASP_YSTORE", "DASP_Y_WT"): os.environ[k] = "0"
    for k, v in kw.items(): os.environ[k] = str(v)
for name in (sys.argv[1:] or ["HV15R", "Queen_4147"]):
    rows, cols = D.synth_dims(name, 1.0)
    rp, ci = D.synth_csr(name, 1.0)
    rng = np.random.default_rng(5)
    plans = [D.Plan(rp, ci, rng.uniform(-1, 1, ci.size), cols).upload() for _ in range(2)]
    for p in plans: p.drop_host()
    del ci
    x = torch.rand(cols, dtype=torch.float64, device="cuda")
    ys = [torch.zeros(rows, dtype=torch.float64, device="cuda") for _ in range(2)]
    env(); plans[0].spmv(x.data_ptr(), ys[0].data_ptr()); torch.cuda.synchronize(); want = ys[0].clone()
    for per_cu in (6, -6, 75):
        env(**({"DASP_HYBRID": per_cu} if per_cu > 6 else {"DASP_PERSIST": per_cu} if per_cu > 0 else {"DASP_PERSIST2": -per_cu})); ys[1].zero_(); plans[0].spmv(x.data_ptr(), ys[1].data_ptr()); torch.cuda.synchronize()
        print(name, "persist", per_cu, "bit-equal to the product kernel:", bool(torch.equal(want, ys[1])), flush=True)
    for k, p in enumerate(plans):
        for j, y in enumerate(ys):
            t = lambda: p.time(x.data_ptr(), y.data_ptr(), 0, 10, 100)[1]
            env(); line = "%-11s plan %d y %d: product %.4f |" % (name, k, j, t())
            cases = [("kt plain", dict(DASP_KT_KERNEL=1)), ("kt sc1", dict(DASP_KT_KERNEL=1, DASP_YSTORE=10)), ("kt nostore", dict(DASP_KT_KERNEL=1, DASP_YSTORE=3)), ("p6 plain", dict(DASP_PERSIST=6))]
            if os.environ.get("PROBE_ALL"):
                cases += [("p6 sc1", dict(DASP_PERSIST=6, DASP_YSTORE=10)), ("p6 nostore", dict(DASP_PERSIST=6, DASP_YSTORE=3)), ("p5 plain", dict(DASP_PERSIST=5)),
                          ("r1 nostore", dict(DASP_KT_REPS=1, DASP_YSTORE=3)), ("r2", dict(DASP_KT_REPS=2, DASP_YSTORE=3)), ("r4", dict(DASP_KT_REPS=4, DASP_YSTORE=3)), ("r8", dict(DASP_KT_REPS=8, DASP_YSTORE=3)),
                          ("d6/64 plain", dict(DASP_PERSIST2=6)), ("d6/256", dict(DASP_PERSIST2=6, DASP_PERSIST_GROUPS=256))]
            for pct in (50, 65, 75, 85, 92):      # hybrid grid: share of the medium workgroups in the resident head
                cases += [("h%d" % pct, dict(DASP_HYBRID=pct)), ("h%d sc1" % pct, dict(DASP_HYBRID=pct, DASP_YSTORE=10))]
            cases += [("h75/5", dict(DASP_HYBRID=75, DASP_HYBRID_PER_CU=5)), ("h75 nostore", dict(DASP_HYBRID=75, DASP_YSTORE=3)), ("kt plain again", dict(DASP_KT_KERNEL=1))]
            for label, kw in cases:
                env(**kw); line += " %s %.4f" % (label, t())
            print(line, flush=True)
    for p in plans: p.close()
    del x, ys

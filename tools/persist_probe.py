#!/usr/bin/env python3
"""tools/persist_probe.py (experiment build): resident workgroups that keep their medium blocks' results in LDS and write y in one burst
(dasp_spmv_persist_kernel) against one block per wave with direct stores (dasp_spmv_kt_kernel), per (plan, y) allocation pair."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import dasp_amd as D
def env(**kw):
    for k in ("DASP_KT_KERNEL", "DASP_PERSIST", "DASP_PERSIST2", "DASP_PERSIST_GROUPS", "DASP_KT_REPS", "DASP_YSTORE", "DASP_Y_WT"): os.environ[k] = "0"
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
    for per_cu in (6, 4, -6):
        env(**({"DASP_PERSIST": per_cu} if per_cu > 0 else {"DASP_PERSIST2": -per_cu})); ys[1].zero_(); plans[0].spmv(x.data_ptr(), ys[1].data_ptr()); torch.cuda.synchronize()
        print(name, "persist", per_cu, "bit-equal to the product kernel:", bool(torch.equal(want, ys[1])), flush=True)
    for k, p in enumerate(plans):
        for j, y in enumerate(ys):
            t = lambda: p.time(x.data_ptr(), y.data_ptr(), 0, 10, 100)[1]
            env(); line = "%-11s plan %d y %d: product %.4f |" % (name, k, j, t())
            for label, kw in (("kt plain", dict(DASP_KT_KERNEL=1)), ("kt sc1", dict(DASP_KT_KERNEL=1, DASP_YSTORE=10)), ("kt nostore", dict(DASP_KT_KERNEL=1, DASP_YSTORE=3)),
                              ("p6 plain", dict(DASP_PERSIST=6)), ("p6 sc1", dict(DASP_PERSIST=6, DASP_YSTORE=10)), ("p6 nt", dict(DASP_PERSIST=6, DASP_YSTORE=13)), ("p6 nostore", dict(DASP_PERSIST=6, DASP_YSTORE=3)),
                              ("p5 plain", dict(DASP_PERSIST=5)),
                              ("r1 nostore", dict(DASP_KT_REPS=1, DASP_YSTORE=3)), ("r2", dict(DASP_KT_REPS=2, DASP_YSTORE=3)), ("r3", dict(DASP_KT_REPS=3, DASP_YSTORE=3)), ("r4", dict(DASP_KT_REPS=4, DASP_YSTORE=3)),
                              ("r8", dict(DASP_KT_REPS=8, DASP_YSTORE=3)),
                              ("d6/64 plain", dict(DASP_PERSIST2=6)), ("d6/64 sc1", dict(DASP_PERSIST2=6, DASP_YSTORE=10)), ("d6/64 nostore", dict(DASP_PERSIST2=6, DASP_YSTORE=3)),
                              ("d6/8", dict(DASP_PERSIST2=6, DASP_PERSIST_GROUPS=8)), ("d6/32", dict(DASP_PERSIST2=6, DASP_PERSIST_GROUPS=32)), ("d6/256", dict(DASP_PERSIST2=6, DASP_PERSIST_GROUPS=256)), ("d6/768", dict(DASP_PERSIST2=6, DASP_PERSIST_GROUPS=768)),
                              ("d6/64 cap8", dict(DASP_PERSIST2=6, DASP_PERSIST_CAP=8)), ("kt plain again", dict(DASP_KT_KERNEL=1))):
                env(**kw); line += " %s %.4f" % (label, t())
            print(line, flush=True)
    for p in plans: p.close()
    del x, ys

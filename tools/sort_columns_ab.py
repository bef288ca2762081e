#!/usr/bin/env python3
"""tools/sort_columns_ab.py -- dasp_options_t::sort_columns on rows whose columns arrive in random order (the stand-ins' rows shuffled, seeded): ms per SpMV with the CSR order kept
(the default, what the reference does) against the rows sorted at pack time, on the host and -- the same CSR living on the GPU -- by the segmented sort of dasp_plan_create_device."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import dasp_amd as D
for spec in (sys.argv[1:] or ["rmat_2M:16", "powerlaw_1M:64", "ljournal-2008:16", "webbase-1M:16"]):
    name, prec = spec.split(":"); prec = int(prec)
    rows, cols = D.synth_dims(name, 1.0)
    rp, ci = D.synth_csr(name, 1.0)
    dt, tdt = (np.float64, torch.float64) if prec == 64 else (np.float16, torch.float16)
    rng = np.random.default_rng(11)
    rowid = np.repeat(np.arange(rows, dtype=np.int64), np.diff(rp))
    o = np.lexsort((rng.random(ci.size), rowid))                    # a random order inside every row
    ci = ci[o]; val = rng.uniform(0.5, 1.5, ci.size).astype(dt)
    x = torch.from_numpy(rng.uniform(0.5, 1.5, cols).astype(dt)).cuda(); y = torch.zeros(rows, dtype=tdt, device="cuda")
    line, ys = "%-16s f%d:" % (name, prec), []
    for label, kw in (("CSR order kept", dict()), ("sort_columns=1 (host)", dict(sort_columns=1))):
        t0 = time.time(); p = D.Plan(rp, ci, val, cols, precision=prec, **kw); pre = time.time() - t0
        p.upload(); p.drop_host()
        line += "  %s %.4f ms (build %.2f s)" % (label, p.time(x.data_ptr(), y.data_ptr(), 0, 20, 200)[1], pre); ys.append(y.clone()); p.close()
    d_rp, d_ci, d_v = torch.from_numpy(rp).cuda(), torch.from_numpy(ci).cuda(), torch.from_numpy(val).cuda()
    for label, kw in (("device build, CSR order", dict()), ("device build, sort_columns=1", dict(sort_columns=1))):
        torch.cuda.synchronize(); t0 = time.time()
        p = D.Plan.from_device(d_rp.data_ptr(), d_ci.data_ptr(), d_v.data_ptr(), rows, cols, int(rp[-1]), precision=prec, **kw)
        torch.cuda.synchronize(); pre = time.time() - t0
        line += "  %s %.4f ms (build %.0f ms)" % (label, p.time(x.data_ptr(), y.data_ptr(), 0, 20, 200)[1], pre * 1e3); ys.append(y.clone()); p.close()
    line += "  | sorted host == sorted device: %s; max |sorted - unsorted| / |y| = %.1e" % (bool(torch.equal(ys[1], ys[3])), float(((ys[1].double() - ys[0].double()).abs() / ys[0].double().abs().clamp(min=1)).max()))
    print(line, flush=True)

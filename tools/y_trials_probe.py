#!/usr/bin/env python3
"""tools/y_trials_probe.py [workload=HV15R] [n_y=8] [n_plans=2]: can the two placement speeds be chosen through the WRITTEN VECTOR alone?
`n_plans` uploads of one plan (trials off), `n_y` separately hipMalloc'ed y vectors of rowA doubles (all alive together): the time of every
(plan, y) pair, twice (is a pair's class stable?), and the same with 64-MiB y allocations (does the size of the allocation matter?)."""
import os, sys, ctypes as C
os.environ["DASP_PLACEMENT_TRIALS"] = "1"
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import dasp_amd as D
from dasp_amd.multi import StreamTimer
name = sys.argv[1] if len(sys.argv) > 1 else "HV15R"
n_y = int(sys.argv[2]) if len(sys.argv) > 2 else 8
n_plans = int(sys.argv[3]) if len(sys.argv) > 3 else 2
rows, cols = D.synth_dims(name, 1.0)
rp, ci = D.synth_csr(name, 1.0)
v = np.ones(ci.size, np.float64)
plans = []
for k in range(n_plans):
    p = D.Plan(rp, ci, v, cols, precision=64).upload(); p.drop_host(); plans.append(p)
del ci, v
hip = StreamTimer._runtime()
def dmalloc(nbytes):
    p = C.c_void_p()
    assert hip.hipMalloc(C.byref(p), C.c_size_t(nbytes)) == 0
    assert hip.hipMemset(p, 0, C.c_size_t(nbytes)) == 0
    return p.value
x = torch.ones(cols, dtype=torch.float64, device="cuda")
IT = int(os.environ.get("PROBE_ITERS", "60"))
def t(p, y): return p.time(x.data_ptr(), y, 0, 4, IT)[1]
for label, nbytes in (("y = rowA doubles", (rows + 64) * 8), ("y inside 64-MiB allocations", 64 << 20), ("y inside 4-KiB-odd allocations", (rows + 64) * 8 + 4096 * 37)):
    ys = [dmalloc(nbytes) for _ in range(n_y)]
    torch.cuda.synchronize()
    for rnd in range(2):
        for k, p in enumerate(plans):
            print("%s | %s | round %d plan %d: " % (name, label, rnd, k) + " ".join("%.4f" % t(p, y) for y in ys), flush=True)
    print("   addresses: " + " ".join(hex(y) for y in ys), flush=True)

#!/usr/bin/env python3
"""tools/y_spacer_probe.py [workload=HV15R]: y candidates with SPACER allocations between them -- how far apart (in allocated bytes) must two small
allocations be to fall into different placement classes against one plan?  One plan (trials off); candidate k = hipMalloc(rowA doubles) after a
spacer of SP[k] bytes was allocated (spacers stay alive until the end)."""
import os, sys, ctypes as C
os.environ["DASP_PLACEMENT_TRIALS"] = "1"
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import dasp_amd as D
from dasp_amd.multi import StreamTimer
name = sys.argv[1] if len(sys.argv) > 1 else "HV15R"
rows, cols = D.synth_dims(name, 1.0)
rp, ci = D.synth_csr(name, 1.0)
plans = []
for k in range(2):
    p = D.Plan(rp, ci, np.ones(ci.size), cols, precision=64).upload(); p.drop_host(); plans.append(p)
del ci
hip = StreamTimer._runtime()
def dmalloc(nbytes, touch=True):
    p = C.c_void_p()
    assert hip.hipMalloc(C.byref(p), C.c_size_t(nbytes)) == 0
    if touch: assert hip.hipMemset(p, 0, C.c_size_t(nbytes)) == 0
    return p.value
x = torch.ones(cols, dtype=torch.float64, device="cuda")
def t(p, y): return p.time(x.data_ptr(), y, 0, 4, 60)[1]
MB = 1 << 20
SP = [0, 0, 32 * MB, 32 * MB, 128 * MB, 128 * MB, 512 * MB, 512 * MB, 1024 * MB, 1024 * MB, 2048 * MB, 2048 * MB, 4096 * MB, 4096 * MB, 8192 * MB, 8192 * MB]
tot = 0
for k, sp in enumerate(SP):
    if sp: dmalloc(sp, touch=False); tot += sp
    y = dmalloc((rows + 64) * 8); tot += (rows + 64) * 8
    torch.cuda.synchronize()
    print("%s candidate %2d after %7.0f MB of allocations (spacer %5d MB) at %s: plan0 %.4f plan1 %.4f ms" % (name, k, tot / MB, sp // MB, hex(y), t(plans[0], y), t(plans[1], y)), flush=True)

#!/usr/bin/env python3
"""tools/placement_cure_probe.py [workload=HV15R] [copies=5]: what could CURE the two placement speeds of the HBM-bound kernels (profiles/r03_placement.md)
instead of drawing allocations until a fast one turns up.  One process, the same plan uploaded `copies` times (trials off), then
  1. every plan against two separately allocated x / y pairs (which plans are slow here?),
  2. the slowest and the fastest plan with y (and x) at OFFSETS inside one big allocation (is the property one of the allocation or of address bits?),
  3. (experiment build only, DASP_AMD_SO=dasp_amd/variants/exp/libdasp_amd.so) the same with non-temporal / write-through y stores."""
import os, sys
os.environ["DASP_PLACEMENT_TRIALS"] = "1"
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import dasp_amd as D
name = sys.argv[1] if len(sys.argv) > 1 else "HV15R"
copies = int(sys.argv[2]) if len(sys.argv) > 2 else 5
rows, cols = D.synth_dims(name, 1.0)
rp, ci = D.synth_csr(name, 1.0)
v = np.ones(ci.size, np.float64)
plans = []
for k in range(copies):
    if os.environ.get("PROBE_XCD_LAST") and k == copies - 1: os.environ["DASP_XCD_BLOCKS"] = "1"      # the last copy deals its blocks to the XCDs in contiguous ranges
    p = D.Plan(rp, ci, v, cols, precision=64).upload(); p.drop_host(); plans.append(p)
os.environ.pop("DASP_XCD_BLOCKS", None)
del ci, v
IT = int(os.environ.get("PROBE_ITERS", "100"))
def t(p, x, y): return p.time(x, y, 0, 5, IT)[1]
xs = [torch.ones(cols, dtype=torch.float64, device="cuda") for _ in range(2)]
ys = [torch.zeros(rows + 64, dtype=torch.float64, device="cuda") for _ in range(2)]
tab = np.zeros((copies, 2))
for rnd in range(2):
    for k, p in enumerate(plans):
        for j in range(2): tab[k, j] += t(p, xs[j].data_ptr(), ys[j].data_ptr()) / 2
for k in range(copies): print("plan %d: pair0 %.4f pair1 %.4f ms" % (k, tab[k, 0], tab[k, 1]), flush=True)
slow, fast = int(np.argmax(tab.min(axis=1))), int(np.argmin(tab.min(axis=1)))
print("slowest plan %d, fastest plan %d" % (slow, fast), flush=True)
big = torch.zeros((3 << 30) // 8, dtype=torch.float64, device="cuda")          # 3 GiB: y / x at chosen offsets inside ONE allocation
bigx = torch.ones((1 << 30) // 8, dtype=torch.float64, device="cuda")
offs = [0, 4 << 10, 64 << 10, 1 << 20, 2 << 20, 16 << 20, 128 << 20, 512 << 20, 1 << 30, 2 << 30]
for which, k in (("slow", slow), ("fast", fast)):
    p = plans[k]
    print("%s plan %d, y at offsets of one 3-GiB allocation (x = pair 0): " % (which, k) + "  ".join("%s:%.4f" % (hex(o), t(p, xs[0].data_ptr(), big.data_ptr() + o)) for o in offs), flush=True)
    print("%s plan %d, x at offsets of one 1-GiB allocation (y = pair 0): " % (which, k) + "  ".join("%s:%.4f" % (hex(o), t(p, bigx.data_ptr() + o, ys[0].data_ptr())) for o in offs if o + cols * 8 <= (1 << 30)), flush=True)
if "variants/exp" in os.environ.get("DASP_AMD_SO", ""):
    for which, k in (("slow", slow), ("fast", fast)) + ((("xcd-contiguous", copies - 1),) if os.environ.get("PROBE_XCD_LAST") else ()):
        for j in range(2):
            line = "%s plan %d pair %d:" % (which, k, j)
            for mode, label in ((0, "plain"), (1, "nt"), (2, "write-through"), (3, "no y store"), (4, "y into 4 KiB"), (0, "plain again")):
                os.environ["DASP_YSTORE"] = str(mode)
                line += "  %s %.4f" % (label, t(plans[k], xs[j].data_ptr(), ys[j].data_ptr()))
            print(line, flush=True)
    os.environ["DASP_YSTORE"] = "0"

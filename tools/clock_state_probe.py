#!/usr/bin/env python3
"""tools/clock_state_probe.py [workload=HV15R]: does the HBM-bound kernel's speed depend on what the GPU did just before?  Back-to-back launches in chunks of 200
(~90 ms each) for several seconds, after (a) a long idle pause, (b) a CPU-heavy phase like plan building; prints ms per launch per chunk over time."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import dasp_amd as D
name = sys.argv[1] if len(sys.argv) > 1 else "HV15R"
rows, cols = D.synth_dims(name, 1.0)
rp, ci = D.synth_csr(name, 1.0)
p = D.Plan(rp, ci, np.ones(ci.size), cols).upload(); p.drop_host()
x = torch.ones(cols, dtype=torch.float64, device="cuda"); y = torch.zeros(rows, dtype=torch.float64, device="cuda")


def series(tag, chunks):
    out = []
    t0 = time.time()
    for c in range(chunks):
        out.append(p.time(x.data_ptr(), y.data_ptr(), 0, 0, 200)[1])
    print("%-34s %s  (%.1f s)" % (tag, " ".join("%.4f" % v for v in out), time.time() - t0), flush=True)


series("right after the plan build:", 24)
time.sleep(10.0)
series("after 10 s idle:", 24)
series("continuing:", 24)
t0 = time.time()
while time.time() - t0 < 5.0:          # CPU-heavy, GPU idle
    np.sort(np.random.default_rng(1).integers(0, 1 << 30, 2_000_000))
series("after 5 s of CPU work, GPU idle:", 24)
for k in range(3):
    time.sleep(1.0)
    series("after 1 s idle:", 8)

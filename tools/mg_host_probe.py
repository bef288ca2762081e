#!/usr/bin/env python3
"""tools/mg_host_probe.py: host-side cost of one multi-GPU step (world size 1, real RCCL communicator): how long the CPU needs to ENQUEUE
plan.spmv / dasp_mg_product / dasp_mg_allgather / dasp_mg_spmv (no sync inside the loop), next to the device time of the same calls."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import dasp_amd as D
from dasp_amd.multi import MgPlan, unique_id
scale = float(sys.argv[1]) if len(sys.argv) > 1 else 0.125
name = "HV15R"
rows, cols = D.synth_dims(name, scale)
rp, ci = D.synth_csr(name, scale)
val = np.repeat(0.5 / np.maximum(np.diff(rp), 1), np.diff(rp))
mg = MgPlan(rp, ci, val, rows, cols, np.array([0, rows], np.int32), 0).upload()
mg.comm_init(unique_id())
mg.set_x(np.ones(cols))
print("stream_memops =", mg.info["stream_memops"], flush=True)
s = torch.cuda.current_stream().cuda_stream
own = mg.subplan(0)
x = torch.ones(own.x_len, dtype=torch.float64, device="cuda"); y = torch.zeros(mg.stride, dtype=torch.float64, device="cuda")
def bench(tag, f, n=300):
    for _ in range(20): f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter(); e0.record()
    for _ in range(n): f()
    t1 = time.perf_counter(); e1.record()
    torch.cuda.synchronize()
    print("%-28s host enqueue %.1f us / call, device %.1f us / call" % (tag, (t1 - t0) / n * 1e6, e0.elapsed_time(e1) / n * 1e3), flush=True)
bench("plan.spmv (own plan)", lambda: own.spmv(x.data_ptr(), y.data_ptr(), s))
bench("mg.product", lambda: mg.product(s))
bench("mg.allgather", lambda: mg.allgather(s))
bench("mg.spmv (product+allgather)", lambda: mg.spmv(s))
mg.wait(s); torch.cuda.synchronize()

#!/usr/bin/env python3
"""tools/mg_overlap_probe.py [n_gpus=8] [rank=3] [workload=HV15R]: ONE rank's share of an n-way row partition on this box's single GPU, with the real
plans (own / other columns), the real two-stream choreography of dasp_mg_spmv and an EMULATED all-gather (test hook DASP_MG_FAKE_ALLGATHER_US: a local copy
plus a kernel that holds the communication stream for that long).  Shows on hardware what the overlap buys: step time vs all-gather duration, with and
without the own / other split."""
import os, subprocess, sys
if len(sys.argv) > 1 and sys.argv[1] == "--child":
    import numpy as np, torch
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import dasp_amd as D
    from dasp_amd.multi import MgPlan
    world, rank, name, overlap = int(sys.argv[2]), int(sys.argv[3]), sys.argv[4], sys.argv[5] == "1"
    rows, cols = D.synth_dims(name, 1.0)
    lengths = D.synth_row_lengths(name, 1.0)
    rpf = np.zeros(rows + 1, np.int64); np.cumsum(lengths, out=rpf[1:])
    bounds = np.searchsorted(rpf, rpf[-1] * np.arange(world + 1) // world, side="left").astype(np.int32)
    bounds[0], bounds[-1] = 0, rows
    r0, r1 = int(bounds[rank]), int(bounds[rank + 1])
    rp, ci = D.synth_csr(name, 1.0, r0, r1, lengths=lengths[r0:r1])
    val = np.repeat(0.5 / np.maximum(np.diff(rp), 1), np.diff(rp))
    mg = MgPlan(rp, ci, val, rows, cols, bounds, rank, overlap=overlap).upload()
    mg.set_x(np.ones(cols))
    s = torch.cuda.current_stream().cuda_stream
    for _ in range(20): mg.spmv(s)
    mg.wait(s); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    n = 300
    e0.record()
    for _ in range(n): mg.spmv(s)
    mg.wait(s); e1.record(); torch.cuda.synchronize()
    own = mg.subplan(0); oth = mg.subplan(1)
    x = torch.ones(own.x_len, dtype=torch.float64, device="cuda"); y = torch.zeros(mg.stride, dtype=torch.float64, device="cuda")
    t_own = own.time(x.data_ptr(), y.data_ptr(), s, warmup=5, iters=100)[1]
    t_oth = 0.0
    if oth is not None:
        xo = torch.ones(oth.x_len, dtype=torch.float64, device="cuda")
        t_oth = oth.time(xo.data_ptr(), y.data_ptr(), s, warmup=5, iters=100)[1]
    print("fake all-gather %3s us  overlap=%d  memops=%d | step %.1f us | own-column product alone %.1f us (%d nnz), other-column %.1f us (%d nnz)" %
          (os.environ.get("DASP_MG_FAKE_ALLGATHER_US"), overlap, mg.info["stream_memops"], e0.elapsed_time(e1) / n * 1e3, t_own * 1e3, mg.nnz_local, t_oth * 1e3, mg.nnz_remote), flush=True)
    sys.exit(0)
world = sys.argv[1] if len(sys.argv) > 1 else "8"
rank = sys.argv[2] if len(sys.argv) > 2 else "3"
name = sys.argv[3] if len(sys.argv) > 3 else "HV15R"
for overlap in ("1", "0"):
    for us in ("0", "20", "40", "60", "80"):
        env = dict(os.environ, DASP_MG_FAKE_ALLGATHER_US=us)
        r = subprocess.run([sys.executable, os.path.abspath(__file__), "--child", world, rank, name, overlap], env=env, capture_output=True, text=True)
        print("\n".join(l for l in (r.stdout + r.stderr).splitlines() if "fake all-gather" in l or "Error" in l or "error" in l), flush=True)

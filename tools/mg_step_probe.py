#!/usr/bin/env python3
"""tools/mg_step_probe.py [n_gpus=8] [workload=HV15R] [ranks=all] [balance=nnz|cost]: EVERY rank's share of an n-way row partition, one after the
other on this box's single GPU, with the real plans and the real choreography of dasp_mg_spmv and an EMULATED exchange (dasp_mg_set_fake_exchange: local
copy + a kernel holding the communication stream).  Per rank: the step time of the fused one-launch form and of the two-launch form for several exchange
durations, the products alone, and -- last line -- the max over ranks against the 1-GPU step of the same box."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import dasp_amd as D
from dasp_amd.multi import MgPlan, StreamTimer

world = int(sys.argv[1]) if len(sys.argv) > 1 else 8
name = sys.argv[2] if len(sys.argv) > 2 else "HV15R"
ranks = list(range(world)) if len(sys.argv) <= 3 or sys.argv[3] == "all" else [int(r) for r in sys.argv[3].split(",")]
balance = sys.argv[4] if len(sys.argv) > 4 else "nnz"
scale = float(os.environ.get("PROBE_SCALE", "1.0"))
AG = [int(a) for a in os.environ.get("PROBE_AG_US", "0,40,60").split(",")]
rows, cols = D.synth_dims(name, scale)
lengths = D.synth_row_lengths(name, scale)
rpf = np.zeros(rows + 1, np.int64); np.cumsum(lengths, out=rpf[1:])
if balance == "cost":
    bounds = D.partition_rows_cost(lengths, world, 64)
else:
    bounds = np.searchsorted(rpf, rpf[-1] * np.arange(world + 1) // world, side="left").astype(np.int32)
    bounds[0], bounds[-1] = 0, rows
s = torch.cuda.current_stream().cuda_stream
s0 = s

HOST = {}
def step_time(mg, n=300):
    for _ in range(20): mg.spmv(s)
    mg.wait(s); torch.cuda.synchronize()
    t = StreamTimer(s)
    t.start(); h0 = time.perf_counter()
    for _ in range(n): mg.spmv(s)
    h1 = time.perf_counter()
    mg.wait(s); ms = t.stop(); torch.cuda.synchronize()
    mg.check()
    HOST["us"] = (h1 - h0) / n * 1e6
    return ms / n * 1e3

worst = {}
for rank in ranks:
    r0, r1 = int(bounds[rank]), int(bounds[rank + 1])
    TR = (lambda what: (sys.stderr.write("rank %d: %s\n" % (rank, what)), sys.stderr.flush())) if os.environ.get("PROBE_TRACE") else (lambda what: None)
    TR("generate")
    rp, ci = D.synth_csr(name, scale, r0, r1, lengths=lengths[r0:r1])
    val = np.repeat(0.5 / np.maximum(np.diff(rp), 1), np.diff(rp))
    TR("create")
    OVL = int(os.environ.get("PROBE_OVERLAP", "1"))          # 2: ONE plan, the step on one stream (needs PROBE_EXCHANGE=push)
    mg = MgPlan(rp, ci, val, rows, cols, bounds, rank, overlap=OVL)
    TR("upload")
    mg.upload()
    del ci, val
    TR("first set_x")
    mg.set_x(np.ones(cols))
    TR("ready")
    line = "rank %d rows %d nnz own %d other %d fused_ok %d |" % (rank, r1 - r0, mg.nnz_local, mg.nnz_remote, mg.info["fused_step"])
    can_fuse = mg.info["fused_step"] >= 1
    if os.environ.get("PROBE_RESERVE"):                  # the plan's CU-masked compute stream (what the RCCL exchange needs)
        s = mg.reserved_stream(int(os.environ["PROBE_RESERVE"]))
        assert s, "no reserved stream"
    if os.environ.get("PROBE_EXCHANGE") == "push":      # the direct exchange, scratch memory standing in for the peers (+ the emulated link time)
        mg.push_loopback()
    if OVL == 2: can_fuse = mg.info["fused_step"] == 2
    for fused in ([] if os.environ.get("PROBE_KERNEL_ONLY") == "1" else [True] if OVL == 2 else [True, False] if can_fuse else [False]):
        mg.set_fused(fused)
        for us in AG:
            mg.set_fake_exchange(us)
            if os.environ.get("PROBE_TRACE"): sys.stderr.write("rank %d fused %d exchange %d us: set_x\n" % (rank, fused, us)); sys.stderr.flush()
            mg.set_x(np.ones(cols))
            if os.environ.get("PROBE_TRACE"): sys.stderr.write("rank %d fused %d exchange %d us: steps\n" % (rank, fused, us)); sys.stderr.flush()
            t = step_time(mg)
            key = ("fused" if fused else "2launch", us)
            worst[key] = max(worst.get(key, 0.0), t)
            line += " %s/%dus %.1f (host %.1f)" % (key[0], us, t, HOST["us"])
    if os.environ.get("PROBE_TRACE"): sys.stderr.write("rank %d: parts alone\n" % rank); sys.stderr.flush()
    if can_fuse:      # the step kernel alone, back to back (no exchange, no in-kernel wait)
        mg.set_fused(True); mg.set_x(np.ones(cols))
        for _ in range(10): mg.product(s)
        torch.cuda.synchronize()
        t = StreamTimer(s)
        t.start(); h0 = time.perf_counter()
        for _ in range(100): mg.product(s)
        h1 = time.perf_counter(); ms = t.stop(); torch.cuda.synchronize()
        line += " | step kernel alone %.1f (host enqueue %.1f)" % (ms / 100 * 1e3, (h1 - h0) * 1e4)
    own = mg.subplan(0); oth = mg.subplan(1)
    x = torch.ones(own.x_len, dtype=torch.float64, device="cuda"); y = torch.zeros(mg.stride, dtype=torch.float64, device="cuda")
    if OVL == 2:
        if "variants/exp" in os.environ.get("DASP_AMD_SO", ""):
            os.environ["DASP_MG_STEP2_NOPUSH"] = "1"
            mg.set_x(np.ones(cols))
            for _ in range(10): mg.product(s)
            torch.cuda.synchronize()
            t = StreamTimer(s); t.start()
            for _ in range(100): mg.product(s)
            ms = t.stop(); torch.cuda.synchronize()
            os.environ["DASP_MG_STEP2_ALLFREE"] = "1"
            for _ in range(10): mg.product(s)
            torch.cuda.synchronize()
            t = StreamTimer(s); t.start()
            for _ in range(100): mg.product(s)
            ms2 = t.stop(); torch.cuda.synchronize()
            line += " | ... and every workgroup dispatched at once %.1f" % (ms2 / 100 * 1e3)
            for fm, what in ((99, "the plan's own order, no tables"), (98, "block order table only")):
                os.environ["DASP_MG_STEP2_FENCE"] = str(fm)
                for _ in range(10): mg.product(s)
                torch.cuda.synchronize()
                t = StreamTimer(s); t.start()
                for _ in range(100): mg.product(s)
                ms3 = t.stop(); torch.cuda.synchronize()
                line += " | %s %.1f" % (what, ms3 / 100 * 1e3)
            os.environ.pop("DASP_MG_STEP2_FENCE")
            os.environ["DASP_MG_STEP2_ALLFREE"] = "0"
            os.environ["DASP_MG_STEP2_NOPUSH"] = "0"
            line += " | step kernel without the stores to the peers and without the wait %.1f" % (ms / 100 * 1e3)
            for fm in (0, 1, 3, 2):
                os.environ["DASP_MG_STEP2_FENCE"] = str(fm)
                mg.set_x(np.ones(cols))
                line += " | fence mode %d: %.1f" % (fm, step_time(mg, 200))
            os.environ.pop("DASP_MG_STEP2_FENCE")
        t_own = own.time(x.data_ptr(), y.data_ptr(), s, warmup=5, iters=100)[1] * 1e3
        mg.set_x(np.ones(cols)); mg.product(s); torch.cuda.synchronize()          # k2 = 1: x in half 1, y_local = this rank's slot there
        other_half = mg.gathered_ptr + (-1 if mg.gathered_ptr > mg.x_ptr else 1) * 0
        t_fx = own.time(mg.gathered_ptr, y.data_ptr(), s, warmup=5, iters=100)[1] * 1e3
        t_fy = own.time(x.data_ptr(), mg.y_local_ptr, s, warmup=5, iters=100)[1] * 1e3
        t_fxy = own.time(mg.gathered_ptr, mg.y_local_ptr, s, warmup=5, iters=100)[1] * 1e3       # (x and y alias inside the same half: timing only)
        line += " | plain kernel with fine-grained x %.1f, y %.1f, both %.1f" % (t_fx, t_fy, t_fxy)
        print(line + " | the plan alone (plain kernel, coarse x / y) %.1f us | data_X %.1f MB blocks %d" % (t_own, own.stats["data_X"] / 1e6, own.stats["n_med_blocks"]), flush=True)
        mg.close(); del x, y; torch.cuda.empty_cache()
        continue
    t_own = own.time(x.data_ptr(), y.data_ptr(), s, warmup=5, iters=100)[1] * 1e3
    t_own2 = own.time(mg.y_local_ptr, y.data_ptr(), s, warmup=5, iters=100)[1] * 1e3
    t_own3 = own.time(mg.y_local_ptr, mg.gathered_ptr, s, warmup=5, iters=100)[1] * 1e3
    t_own4 = own.time(mg.gathered_ptr, y.data_ptr(), s, warmup=5, iters=100)[1] * 1e3          # x read from the (fine-grained) gather buffer
    line += " | own alone with x=ys %.1f, and y=yg %.1f, x=yg(fine-grained) %.1f" % (t_own2, t_own3, t_own4)
    if can_fuse and "variants/exp" in os.environ.get("DASP_AMD_SO", ""):      # the own-column part inside the step kernel (no other-column product, no waiting workgroups)
        os.environ["DASP_MG_STEP_NOOTHER"] = "1"
        mg.set_fused(True); mg.set_x(np.ones(cols))
        for _ in range(10): mg.product(s)
        torch.cuda.synchronize()
        t = StreamTimer(s); t.start()
        for _ in range(100): mg.product(s)
        ms = t.stop(); torch.cuda.synchronize()
        os.environ["DASP_MG_STEP_NOOTHER"] = "0"
        line += " | step kernel without the other-column part %.1f" % (ms / 100 * 1e3)
    t_oth = 0.0
    if oth is not None:
        xo = torch.ones(oth.x_len, dtype=torch.float64, device="cuda")
        t_oth = oth.time(xo.data_ptr(), y.data_ptr(), s, warmup=5, iters=100)[1] * 1e3
        del xo
    st = own.stats
    print(line + " | own alone %.1f other alone %.1f us | own data_X %.1f MB blocks %d" % (t_own, t_oth, st["data_X"] / 1e6, st["n_med_blocks"]), flush=True)
    TR("close")
    mg.close(); del x, y
    TR("empty_cache")
    torch.cuda.empty_cache()
    TR("done")
if os.environ.get("PROBE_FULL", "1") == "1":
    rp, ci = D.synth_csr(name, scale, 0, rows, lengths=lengths)
    plan = D.Plan(rp, ci, np.ones(ci.size), cols).upload(); plan.drop_host()
    del ci
    x = torch.ones(cols, dtype=torch.float64, device="cuda"); y = torch.zeros(rows, dtype=torch.float64, device="cuda")
    t1 = plan.time(x.data_ptr(), y.data_ptr(), s0, warmup=10, iters=200)[1] * 1e3
    print("1-GPU step %.1f us" % t1)
    for k in sorted(worst):
        print("max over ranks %-8s exchange %3d us: %.1f us  -> %.2fx" % (k[0], k[1], worst[k], t1 / worst[k]), flush=True)

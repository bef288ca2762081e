#!/usr/bin/env python3
"""bench.py -- DASP SpMV on MI355X: GFLOP/s and achieved fraction of the HBM roofline.

Contract (driver):  python bench.py --gpus N --steps K --warmup W
  N == 1 : one process, one GPU.
  N  > 1 : launched by torch.distributed.run, one rank per GPU (RCCL).  The matrix is partitioned by contiguous
           row ranges of equal nonzero count; every rank builds the DASP plans of its slice (dasp_amd/multi.py): one over
           its own columns, one over the other ranks' columns remapped so that the all-gather buffer IS the next x.
           One step = y = A*x over the whole matrix + all-gather of y over xGMI, chained (x_{t+1} = y_t); the product
           over a rank's own columns overlaps the all-gather still in flight.  Fixed total work => "strong".
A step is one y = A*x over the whole matrix.  Input: the seeded synthetic stand-in of the SuiteSparse matrix named by
--workload (no .mtx files / network on the bench machines).  N == 1: values and x all ones as in the reference's driver
(src/main_f64.cu:131-132), y[i] == nnz(row) checked exactly after the timed region.  N > 1: a_ij = 0.5 / len(row i), x_0 = 1,
so x_t = 2^-t, checked on the gathered y after the timed region.
Prints ONE JSON line on rank 0.  The GPU path has no CPU fallback: without a GPU this exits non-zero.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBPS = 8000.0   # MI355X HBM3E spec peak (MI355X_MICROARCH.md, chip-level parameters)


def algorithmic_bytes(m, n, nnz, vbytes):
    """CSR read once + x read once + y written once: the reference's data_origin1 (main_f64.cu:143)."""
    return (nnz + n + m) * vbytes + nnz * 4 + (m + 1) * 4


_REAL = {}


def real_matrix(D, name):
    """The real SuiteSparse file, when DASP_MTX_DIR holds <name>.mtx; pattern only: values are set to 1
    as the reference's driver does.  Parsed once per process, with a binary CSR cache next to the file."""
    d = os.environ.get("DASP_MTX_DIR")
    if not d:
        return None
    if name not in _REAL:
        path = os.path.join(d, name + ".mtx")
        _REAL[name] = None
        if os.path.exists(path):
            cache = path + ".f64.csrbin"
            try:
                m, n, nnz, sym, rp, ci, v = D.csr_load(cache, 64)
            except Exception:
                m, n, nnz, sym, rp, ci, v = D.mmio_allinone(path, 64)
                try:
                    D.csr_save(cache, rp, ci, v, n, sym, 64)
                except Exception:
                    pass
            _REAL[name] = (m, n, rp.astype(np.int64), ci)
    return _REAL[name]


def matrix_dims(D, name, scale):
    real = real_matrix(D, name)
    return (real[0], real[1]) if real else D.synth_dims(name, scale)


def matrix_lengths(D, name, scale):
    real = real_matrix(D, name)
    return np.diff(real[2]).astype(np.int32) if real else D.synth_row_lengths(name, scale)


def matrix_rows(D, name, scale, r0, r1, lengths):
    real = real_matrix(D, name)
    if real:
        rp, ci = real[2], real[3]
        return (rp[r0:r1 + 1] - rp[r0]).astype(np.int32), np.ascontiguousarray(ci[rp[r0]:rp[r1]])
    return D.synth_csr(name, scale, r0, r1, lengths=lengths[r0:r1])


def build_slice(D, name, scale, precision, r0, r1, lengths, bounds=None, stride=0, natural=False, threads=0):
    rp, ci = matrix_rows(D, name, scale, r0, r1, lengths)
    rows, cols = matrix_dims(D, name, scale)
    dt = np.float64 if precision == 64 else np.float16
    val = np.ones(ci.size, dt)                                  # initVec(csrValA): utils.h:93-100
    t0 = time.time()
    plan = D.Plan(rp, ci, val, cols, precision=precision, y_order=D.Y_NATURAL if natural else D.Y_PERMUTED,
                  part_bounds=bounds, part_stride=stride, host_threads=threads)
    pre_s = time.time() - t0
    return plan, rp, ci, val, pre_s


def time_plan(torch, plan, x, y, iters, warmup):
    s = torch.cuda.current_stream().cuda_stream
    return plan.time(x.data_ptr(), y.data_ptr(), s, warmup=warmup, iters=iters)


def f16_close(torch, got, want):
    """f16 check of the all-ones mode: within 1e-2 of the row length, or +inf where the row length itself exceeds binary16 (65504)"""
    over = want > 65504.0
    fine = (got - want).abs() <= 1e-2 * want.clamp(min=1)
    return bool((torch.where(over, torch.isinf(got) | fine, fine)).all().item())


def suite_entry(torch, D, name, precision, scale, budget_s=2.0):
    """Reference protocol (100 warm-up + up to 1000 timed launches, dasp_f64.h:1285-1286) on one stand-in."""
    rows, cols = matrix_dims(D, name, scale)
    lengths = matrix_lengths(D, name, scale)
    plan, rp, ci, val, pre_s = build_slice(D, name, scale, precision, 0, rows, lengths)
    nnz = int(rp[-1])
    del ci, val
    plan.upload()
    plan.drop_host()
    tdt = torch.float64 if precision == 64 else torch.float16
    x = torch.ones(cols, dtype=tdt, device="cuda")
    y = torch.zeros(rows, dtype=tdt, device="cuda")
    w, e = time_plan(torch, plan, x, y, 20, 10)
    iters = int(max(20, min(1000, budget_s * 1e3 / max(e, 1e-4))))
    w, e = time_plan(torch, plan, x, y, iters, min(100, iters))
    order = torch.from_numpy(plan.order_rid.astype(np.int64)).cuda()
    want = torch.from_numpy(np.diff(rp).astype(np.float64)).cuda()[order]
    ok = bool((y.double() == want).all().item()) if precision == 64 or int(np.diff(rp).max()) <= 2048 else \
        f16_close(torch, y.double(), want)
    st = plan.stats
    b_alg = algorithmic_bytes(rows, cols, nnz, precision // 8)
    out = {"workload": name, "dtype": "f64" if precision == 64 else "f16", "rows": rows, "nnz": nnz,
           "ms": round(w, 6), "event_ms": round(e, 6), "iters": iters, "gflops": round(2.0 * nnz / (w * 1e6), 2),
           "achieved_GBps": round(b_alg / (e * 1e6), 1), "frac_hbm_roofline": round(b_alg / (e * 1e6) / HBM_PEAK_GBPS, 4),
           "rate_fill0": round(st["rate_fill0"], 4), "pre_ms": round(st["pre_ms"], 1), "verified": ok,
           "col_panels": st["n_col_panels"], "row_long": st["row_long"], "row_block": st["row_block"],
           "row_short": rows - st["row_long"] - st["row_block"] - st["row_zero"]}
    plan.close()
    del x, y
    torch.cuda.empty_cache()
    return out


def setup_rank(torch, D, name, scale, prec, rank, world, multi=None, chain=None):
    """Everything one rank owns.  Single GPU: the plan of the whole matrix (A = 1, x = 1, the reference driver's mode).
    Partitioned: its row range (equal nonzeros) as a dasp_amd.multi.RowPartitionedSpMV -- a plan over the rank's own columns
    and one over the other ranks' columns remapped into the all-gather layout, the padded y slices and the gather buffer."""
    multi = world > 1 if multi is None else multi                   # the partitioned layout (forced at world 1 by a test hook)
    rows, cols = matrix_dims(D, name, scale)
    lengths = matrix_lengths(D, name, scale)                        # every rank: cheap, deterministic
    rp_full = np.zeros(rows + 1, np.int64)
    np.cumsum(lengths, out=rp_full[1:])
    nnz_total = int(rp_full[-1])
    if multi:
        # same rule as dasp_partition_rows (first row whose start >= g/world of the nonzeros), on int64 prefix sums
        bounds = np.searchsorted(rp_full, nnz_total * np.arange(world + 1) // world, side="left").astype(np.int32)
        bounds[0], bounds[-1] = 0, rows
        bounds = np.maximum.accumulate(bounds)
        stride = (int(np.diff(bounds).max()) + 63) // 64 * 64
    else:
        bounds, stride = None, 0
    r0, r1 = (0, rows) if not multi else (int(bounds[rank]), int(bounds[rank + 1]))
    threads = max(1, (os.cpu_count() or 8) // max(1, world))
    if multi:
        # chained iteration x_{t+1} = all_gather(A x_t) (what a solver does with the gathered y).  Values c / len(row) make A
        # row-stochastic up to the factor c, so x_t = c^t * ones: bounded for any number of steps, and (f64, c = 1/2, exact
        # powers of two) a product that read a stale x is off by a factor 2 and fails the check.
        from dasp_amd.multi import RowPartitionedSpMV
        rp, ci = matrix_rows(D, name, scale, r0, r1, lengths)
        dt = np.float64 if prec == 64 else np.float16
        c = CHAIN_FACTOR[prec] if chain is None else chain
        val = np.repeat(c / np.maximum(np.diff(rp), 1), np.diff(rp)).astype(dt)
        t0 = time.time()
        mp = RowPartitionedSpMV(torch, rp, ci, val, cols, bounds, rank, precision=prec, threads=threads, stride=stride,
                                overlap=os.environ.get("DASP_BENCH_OVERLAP", "1") != "0")
        pre_s = time.time() - t0
        mp.seed(np.ones(cols, dt))
        del val
        return dict(chain=c, mp=mp, plan=mp.plan, rp=rp, ci=ci, stats=mp.plan.stats, pre_s=pre_s, rows=rows, cols=cols, nnz_total=nnz_total,
                    lengths=lengths, bounds=bounds, stride=stride, r0=r0, r1=r1, x=mp.ys[0], y=mp.ys[1], gathered=mp.gathered)
    plan, rp, ci, val, pre_s = build_slice(D, name, scale, prec, r0, r1, lengths, threads=threads)
    del val
    plan.upload()
    plan.drop_host()
    tdt = torch.float64 if prec == 64 else torch.float16
    x = torch.ones(plan.x_len, dtype=tdt, device="cuda")
    y = torch.zeros(r1 - r0, dtype=tdt, device="cuda")
    return dict(chain=None, mp=None, plan=plan, rp=rp, ci=ci, stats=plan.stats, pre_s=pre_s, rows=rows, cols=cols, nnz_total=nnz_total, lengths=lengths,
                bounds=bounds, stride=stride, r0=r0, r1=r1, x=x, y=y, gathered=None)


CHAIN_FACTOR = {64: 0.5, 16: 1.0}     # f16: 0.5^t would underflow after 24 steps (f64: after 1022, see main)


def cpu_baseline(O, rp, ci, n_cols, budget_s=20.0):
    """Serial CSR SpMV (oracle/dasp_oracle.c, 1 thread) on the same CSR and x: a reported baseline."""
    nnz = int(rp[-1])
    val = np.ones(nnz, np.float64)
    x = np.ones(n_cols, np.float64)
    t = []
    O.csr_spmv(rp, ci, val, x)
    t0 = time.time()
    while len(t) < 3 or (time.time() - t0 < budget_s and len(t) < 15):
        a = time.perf_counter()
        y = O.csr_spmv(rp, ci, val, x)
        t.append(time.perf_counter() - a)
    assert (y == np.diff(rp)).all()
    med = float(np.median(t))
    return {"value": round(2.0 * nnz / med / 1e9, 3), "unit": "GFLOP/s", "cores": 1, "kind": "port",
            "sample": "serial CSR loop over the full %d-row / %d-nnz workload matrix, median of %d passes" % (rp.size - 1, nnz, len(t)),
            "ms": round(med * 1e3, 3), "host_cores_available": os.cpu_count(),
            "achieved_GBps": round(algorithmic_bytes(rp.size - 1, n_cols, nnz, 8) / med / 1e9, 2)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--workload", default="HV15R")
    ap.add_argument("--scale", type=float, default=1.0)
    ap.add_argument("--precision", type=int, default=64, choices=[64, 16])
    ap.add_argument("--no-suite", action="store_true")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-vendor", action="store_true")
    ap.add_argument("--suite-scale", type=float, default=1.0)
    args = ap.parse_args()

    import torch
    import dasp_amd as D

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != max(args.gpus, 1):
        if world == 1 and args.gpus > 1:
            sys.exit("bench.py --gpus %d must be launched with torch.distributed.run --nproc-per-node %d" % (args.gpus, args.gpus))
    if not torch.cuda.is_available():
        sys.exit("bench.py: no GPU visible; the DASP path has no CPU fallback")
    # test hooks (tests/test_gpu_spmv.py runs the N > 1 flow on a one-GPU box): DASP_BENCH_SHARE_GPU=1 puts every rank on
    # cuda:0, DASP_BENCH_BACKEND=gloo stages the all-gather through host memory.  The driver's runs use neither.
    share_gpu = os.environ.get("DASP_BENCH_SHARE_GPU") == "1"
    backend = os.environ.get("DASP_BENCH_BACKEND", "nccl")
    torch.cuda.set_device(0 if share_gpu else local_rank)
    # DASP_BENCH_FORCE_DIST=1 (test hook): run the partitioned + RCCL flow even at world size 1, which is all a one-GPU box
    # can offer RCCL (two ranks may not share a device)
    multi = world > 1 or os.environ.get("DASP_BENCH_FORCE_DIST") == "1"
    dist = None
    if multi:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend)

    name, scale, prec = args.workload, args.scale, args.precision
    vb = prec // 8
    # x_t = c^t must stay a normal number over warmup + steps products: c = 1/2 up to 1000 of them, else 1
    chain = None if args.warmup + args.steps <= 1000 else 1.0
    R = setup_rank(torch, D, name, scale, prec, rank, world, multi, chain)
    plan, rp, ci, st, pre_s = R["plan"], R["rp"], R["ci"], R["stats"], R["pre_s"]
    rows, cols, nnz_total, lengths = R["rows"], R["cols"], R["nnz_total"], R["lengths"]
    bounds, stride, r0, r1, x, y, gathered = R["bounds"], R["stride"], R["r0"], R["r1"], R["x"], R["y"], R["gathered"]
    stream = torch.cuda.current_stream().cuda_stream

    mp = R["mp"]

    class _Done:                                                  # a host-staged exchange has completed when it returns
        def wait(self):
            pass

    def exchange(dst, src):
        if backend == "nccl":                                     # RCCL over xGMI; `dst` has the layout the next product reads
            return dist.all_gather_into_tensor(dst, src, async_op=True)
        parts = [torch.empty(stride, dtype=src.dtype) for _ in range(world)]      # test hook: the same exchange through host memory
        dist.all_gather(parts, src.cpu())
        dst.copy_(torch.cat(parts))
        return _Done()

    def step():
        if mp is None:
            plan.spmv(x.data_ptr(), y.data_ptr(), stream)
        else:
            mp.step(exchange)     # local-column product | wait for the previous all-gather | remote-column product | start the next

    def fence():
        if mp is not None:
            mp.finish()
        torch.cuda.synchronize()
        if multi:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    fence()
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter()
    ev0.record()
    for _ in range(args.steps):
        step()
    ev1.record()
    fence()
    elapsed = time.perf_counter() - t0
    if multi:
        tt = torch.tensor([elapsed], dtype=torch.float64, device="cuda" if backend == "nccl" else "cpu")
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())
    region_event_ms = ev0.elapsed_time(ev1) / args.steps

    if mp is None:
        # ---- exact check: values and x all ones => y == row length
        want = torch.from_numpy(lengths[r0:r1].astype(np.float64)).cuda()
        got = y[: r1 - r0].double()
        got_nat = torch.empty_like(got)
        got_nat[torch.from_numpy(plan.order_rid.astype(np.int64)).cuda()] = got
        ok = bool((got_nat == want).all().item()) if prec == 64 else f16_close(torch, got_nat, want)
    else:
        # ---- chained check: x_t = c^t on every non-empty row (0 on empty ones) after warmup + steps products, on the gathered y
        t_all = args.warmup + args.steps
        full = mp.full_y().double()
        nonempty = torch.from_numpy((lengths > 0).astype(np.float64)).cuda()
        if prec == 64:
            want = (R["chain"] ** t_all) * nonempty
            ok = bool(((full - want).abs() <= 1e-9 * want).all().item())
        else:
            ok = bool(torch.isfinite(full).all().item() and ((full >= 0.5 * nonempty) & (full <= 2.0)).all().item())
        ok = ok and bool(torch.equal(mp.y_local, mp.gathered[rank * stride: rank * stride + (r1 - r0)]))
        okt = torch.tensor([1 if ok else 0], device="cuda" if backend == "nccl" else "cpu")
        dist.all_reduce(okt, op=dist.ReduceOp.MIN)
        ok = bool(okt.item())

    # ---- dominant kernel alone: HIP events on the launch stream around back-to-back launches
    k_iters = max(20, min(args.steps, 1000))
    kw, ke = plan.time(x.data_ptr(), y.data_ptr(), stream, warmup=5, iters=k_iters)
    # partitioned: the dominant kernel is the rank's local-column plan (its x is the rank's own slice)
    nnz_local = int(rp[-1]) if mp is None else mp.nnz_local
    b_alg_local = algorithmic_bytes(r1 - r0, cols if mp is None else stride, nnz_local, vb)
    b_alg_total = algorithmic_bytes(rows, cols, nnz_total, vb)
    achieved = b_alg_local / (ke * 1e6)
    ms_per_step = elapsed * 1e3 / args.steps
    value = 2.0 * nnz_total / (ms_per_step * 1e6)

    vals_desc = "A=1, x=1" if mp is None else "a_ij = %g/len(row i), x_0 = 1, x_{t+1} = y_t" % R["chain"]
    out = {
        "metric": "SpMV GFLOP/s (f64)" if prec == 64 else "SpMV GFLOP/s (f16)", "value": round(value, 2), "unit": "GFLOP/s",
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(ms_per_step, 6),
        "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
        "dtype": "f64" if prec == 64 else "f16 (f32 accumulate)", "data": "suitesparse" if real_matrix(D, name) else "synthetic",
        "config": {"workload": ("%s from DASP_MTX_DIR, %s" % (name, vals_desc)) if real_matrix(D, name) else
                   "%s synthetic stand-in (seeded; SuiteSparse dims/row statistics), %s" % (name, vals_desc),
                   "rows": rows, "cols": cols, "nnz": nnz_total, "scale": scale,
                   "partition": "single GPU" if not multi else
                   ("row ranges by nnz + RCCL all_gather(y) overlapped with the product over the rank's own columns; x_{t+1} = y_t"
                    if mp.overlap else "row ranges by nnz + RCCL all_gather(y); x_{t+1} = y_t"),
                   **({} if mp is None else {"rank0_nnz_own_columns": mp.nnz_local, "rank0_nnz_other_columns": mp.nnz_remote}),
                   "row_long": st["row_long"], "row_block": st["row_block"], "rate_fill0": round(st["rate_fill0"], 4)},
        "roofline": {"bound": "hbm", "achieved": round(achieved, 1), "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                     "frac": round(achieved / HBM_PEAK_GBPS, 4), "traffic": None,
                     "kernel": "dasp_spmv_kernel<%s>" % ("double" if prec == 64 else "_Float16"),
                     "algorithmic_bytes_per_launch": b_alg_local, "kernel_ms": round(ke, 6),
                     "method": "hipEvent pair on the launch stream around %d back-to-back launches (rank 0 slice)" % k_iters},
        "achieved_GBps_whole_job": round(b_alg_total / (ms_per_step * 1e6), 1),
        "frac_hbm_roofline_whole_job": round(b_alg_total / (ms_per_step * 1e6) / (HBM_PEAK_GBPS * world), 4),
        "region_event_ms_per_step": round(region_event_ms, 6), "verified": ok, "preprocess_s": round(pre_s, 3),
    }

    # HBM bytes per launch from the committed rocprofv3 PMC passes of this same workload (tools/prof.sh;
    # FETCH_SIZE x2 gfx950 correction + WRITE_SIZE), when there is one; PMC cannot be read from inside the run
    tpath = os.path.join(ROOT, "profiles", "traffic.json")
    if world == 1 and os.path.exists(tpath):
        for t in json.load(open(tpath)):
            if t["workload"] == name and t["precision"] == prec and abs(t["scale"] - scale) < 1e-12:
                out["roofline"]["traffic"] = int(t["traffic_bytes"])
                out["roofline"]["traffic_source"] = t["source"]
                out["roofline"]["traffic_over_algorithmic"] = round(t["traffic_bytes"] / b_alg_local, 4)

    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        from oracle import oracle as O      # checker / baseline only; never on the measured path
        out["cpu_baseline"] = cpu_baseline(O, rp, ci, cols)
    del ci
    plan.close()
    del x, y
    torch.cuda.empty_cache()

    if rank == 0 and world == 1 and not args.no_vendor:
        # vendor comparator on the same box and matrix: rocSPARSE CSR SpMV (the reference's cuSPARSE column, main_f64.cu:18-100)
        exe = os.path.join(ROOT, "dasp_amd", "bin", "dasp_rocsparse")
        if prec == 64 and os.path.exists(exe) and not real_matrix(D, name):   # the comparator driver generates the stand-in itself
            import re
            import subprocess
            try:
                r = subprocess.run([exe, name, repr(scale), "100", "10"], capture_output=True, text=True, timeout=600)
                mt = re.search(r"\| ([0-9.]+) ms ([0-9.]+) GFLOP/s", r.stdout)
                if mt:
                    out["rocsparse_csr"] = {"ms": float(mt.group(1)), "gflops": float(mt.group(2)),
                                            "speedup_of_dasp": round(float(mt.group(1)) / ms_per_step, 3)}
            except Exception as exc:
                out["rocsparse_csr"] = {"error": repr(exc)}

    if rank == 0 and world == 1 and not args.no_suite:
        suite = []
        for nm, pr in (("cop20k_A", 64), ("nlpkkt160", 64), ("powerlaw_1M", 64), ("Queen_4147", 64),
                       ("webbase-1M", 16), ("ljournal-2008", 16), ("rmat_2M", 16)):
            try:
                suite.append(suite_entry(torch, D, nm, pr, args.suite_scale))
            except Exception as exc:   # a failing extra must not hide the headline line
                suite.append({"workload": nm, "error": repr(exc)})
        out["suite"] = suite

    if multi:
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0:
        print(json.dumps(out))
    if not ok:
        sys.exit(3)


if __name__ == "__main__":
    main()

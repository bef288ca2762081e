#!/usr/bin/env python3
"""tools/twophase_probe.py [workload=ljournal-2008] [CB=32768] [RB=8192] [splits=8] [threads2=256]
PROTOTYPE driver: the gather-free two-phase f16 SpMV of tools/proto/twophase.hip against the product's plan on the same
full-size stand-in.  Builds the tile-ordered streams with numpy (a stable sort of the nonzeros by (row block, column block)), checks y against a
float64 CSR product at the north_star's f16 tolerance, times phase 1, phase 2 and both, and prints bytes per nonzero and the fraction of the
8 TB/s roofline that B_alg / time gives.  Needs dasp_amd/variants/proto/libtwophase.so (hipcc line in the .hip file)."""
import ctypes, os, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import dasp_amd as D

lib = ctypes.CDLL(os.path.join(ROOT, "dasp_amd/variants/proto/libtwophase.so"))
VP, CI = ctypes.c_void_p, ctypes.c_int
lib.tp_phase1.argtypes = [VP, VP, VP, VP, VP, CI, CI, CI, CI, VP]
lib.tp_phase2.argtypes = [VP, VP, VP, VP, VP, CI, CI, CI, CI, VP]
lib.tp_set_lds.argtypes = [CI, CI]


def gpu_time(fn, warm=5, iters=50):
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / iters


def build(rp, ci, val, m, n, CB, RB):
    """tile-ordered streams.  Tiles (row block r, column block c); a tile's nonzeros in CSR order, padded to whole 64-element segments.
    RB-major arrays (phase 2): val2, lrow2 (+ xs, written by phase 1); CB-major arrays (phase 1): lcol1, dst_seg."""
    t0 = time.time()
    n_cb, n_rb = -(-n // CB), -(-m // RB)
    lens = np.diff(rp)
    rows = np.repeat(np.arange(m, dtype=np.int32), lens)
    tile = (rows // RB).astype(np.int32) * n_cb + (ci // CB).astype(np.int32)          # RB-major tile id
    order = np.argsort(tile, kind="stable")
    cnt = np.bincount(tile, minlength=n_rb * n_cb).astype(np.int64)
    segs = (cnt + 63) // 64
    off2 = np.concatenate([[0], np.cumsum(segs)])                                       # RB-major segment offsets, [tiles + 1]
    segsT = segs.reshape(n_rb, n_cb).T.copy()                                            # [c][r]
    off1T = np.concatenate([[0], np.cumsum(segsT.reshape(-1))])[:-1].reshape(n_cb, n_rb).T.reshape(-1)   # CB-major offset of tile (r, c), indexed by the RB-major id
    total = int(off2[-1])
    start = np.concatenate([[0], np.cumsum(cnt)])[:-1]
    ts = tile[order]
    j = np.arange(ci.size, dtype=np.int64) - start[ts]
    pos2 = off2[ts] * 64 + j
    pos1 = off1T[ts] * 64 + j
    val2 = np.zeros(total * 64, np.float16); lrow2 = np.zeros(total * 64, np.uint16); lcol1 = np.zeros(total * 64, np.uint16)
    val2[pos2] = val[order]
    lrow2[pos2] = (rows[order] % RB).astype(np.uint16)
    lcol1[pos1] = (ci[order] % CB).astype(np.uint16)
    nz = np.nonzero(segs)[0]
    rep = segs[nz]
    t = np.arange(total, dtype=np.int64) - np.repeat(np.concatenate([[0], np.cumsum(rep)])[:-1], rep)
    dst_seg = np.zeros(total, np.int32)
    dst_seg[np.repeat(off1T[nz], rep) + t] = (np.repeat(off2[nz], rep) + t).astype(np.int32)
    cb_seg0 = np.concatenate([[0], np.cumsum(segsT.sum(axis=1))]).astype(np.int32)
    rb_seg0 = off2[::n_cb].astype(np.int32)
    assert rb_seg0.size == n_rb + 1 and cb_seg0.size == n_cb + 1 and cb_seg0[-1] == total
    print("  format: %d x %d tiles, %.0f nonzeros per tile, padding %.3f, built in %.1f s" % (n_rb, n_cb, ci.size / max(1, (cnt > 0).sum()), total * 64 / ci.size - 1, time.time() - t0), flush=True)
    return dict(val2=val2, lrow2=lrow2, lcol1=lcol1, dst_seg=dst_seg, cb_seg0=cb_seg0, rb_seg0=rb_seg0, n_cb=n_cb, n_rb=n_rb, total=total)


def main():
    name = sys.argv[1] if len(sys.argv) > 1 else "ljournal-2008"
    CBs = [int(v) for v in (sys.argv[2] if len(sys.argv) > 2 else "32768").split(",")]
    RBs = [int(v) for v in (sys.argv[3] if len(sys.argv) > 3 else "8192").split(",")]
    splits_l = [int(v) for v in (sys.argv[4] if len(sys.argv) > 4 else "8").split(",")]
    thr2 = [int(v) for v in (sys.argv[5] if len(sys.argv) > 5 else "256").split(",")]
    scale = float(os.environ.get("TP_SCALE", "1.0"))
    m, n = D.synth_dims(name, scale)
    rp, ci = D.synth_csr(name, scale)
    rng = np.random.default_rng(3)
    val = rng.uniform(0.5, 1.5, ci.size).astype(np.float16)
    xh = rng.uniform(0.5, 1.5, n).astype(np.float16)
    nnz = ci.size
    b_alg = nnz * 6 + (m + 1) * 4 + (n + m) * 2
    print(name, "rows", m, "nnz", nnz, "B_alg %.1f MB" % (b_alg / 1e6), flush=True)
    lens = np.diff(rp)
    want = np.add.reduceat(val.astype(np.float64) * xh[ci].astype(np.float64), np.minimum(rp[:-1], nnz - 1)) * (lens > 0)
    scale_r = np.maximum(np.add.reduceat(np.abs(val.astype(np.float64) * xh[ci].astype(np.float64)), np.minimum(rp[:-1], nnz - 1)) * (lens > 0), 1e-300)
    x = torch.from_numpy(xh).cuda()
    y = torch.zeros(m, dtype=torch.float16, device="cuda")
    if os.environ.get("TP_SKIP_PRODUCT") != "1":
        plan = D.Plan(rp, ci, val, n, precision=16, y_order=D.Y_NATURAL).upload()
        plan.drop_host()
        t_prod = gpu_time(lambda: plan.spmv(x.data_ptr(), y.data_ptr()))
        got = y.cpu().numpy().astype(np.float64)
        print("product plan: %.4f ms = %.3f of the roofline; max rel err %.2e" % (t_prod, b_alg / (t_prod * 1e6) / 8000, np.max(np.abs(got - want) / scale_r)), flush=True)
        plan.close()
    assert lib.tp_set_lds(0, 0) == 0
    s = torch.cuda.current_stream().cuda_stream
    for CB in CBs:
        for RB in RBs:
            F = build(rp, ci, val, m, n, CB, RB)
            dv = {k: torch.from_numpy(F[k]).cuda() for k in ("val2", "lcol1", "dst_seg", "cb_seg0", "rb_seg0")}
            dv["lrow2"] = torch.from_numpy(F["lrow2"].view(np.int16)).cuda()
            dv["lcol1"] = torch.from_numpy(F["lcol1"].view(np.int16)).cuda()
            xs = torch.zeros(F["total"] * 64, dtype=torch.float16, device="cuda")
            for splits in splits_l:
                for th in thr2:
                    p1 = lambda: lib.tp_phase1(dv["lcol1"].data_ptr(), dv["dst_seg"].data_ptr(), x.data_ptr(), xs.data_ptr(), dv["cb_seg0"].data_ptr(), CB, n, F["n_cb"], splits, s)
                    p2 = lambda: lib.tp_phase2(dv["val2"].data_ptr(), dv["lrow2"].data_ptr(), xs.data_ptr(), dv["rb_seg0"].data_ptr(), y.data_ptr(), RB, m, F["n_rb"], th, s)
                    y.fill_(float("nan"))
                    assert p1() == 0 and p2() == 0
                    torch.cuda.synchronize()
                    got = y.cpu().numpy().astype(np.float64)
                    err = float(np.max(np.abs(got - want) / scale_r))
                    t1, t2 = gpu_time(p1), gpu_time(p2)
                    tb = gpu_time(lambda: (p1(), p2()))
                    bytes1 = F["total"] * 64 * 4 + F["total"] * 4 + F["n_cb"] * splits * CB * 2
                    bytes2 = F["total"] * 64 * 6 + m * 2
                    print("CB %6d RB %6d splits %2d threads2 %3d | phase 1 %.4f ms (%.2f TB/s) phase 2 %.4f ms (%.2f TB/s) both %.4f ms = %.3f of the roofline | %.2f B/nnz streamed | max rel err %.2e %s"
                          % (CB, RB, splits, th, t1, bytes1 / t1 / 1e9, t2, bytes2 / t2 / 1e9, tb, b_alg / (tb * 1e6) / 8000, (bytes1 + bytes2) / nnz, err, "OK" if err <= 1e-2 else "WRONG"), flush=True)
            del dv, xs
            torch.cuda.empty_cache()


if __name__ == "__main__":
    main()

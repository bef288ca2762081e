#!/usr/bin/env python3
"""tools/stamp_probe.py <workload> <precision> [key=value plan options ...] -- where a small launch spends its time.

Needs the DASP_STAMPS build (tools/build_variant.sh stamps "-DDASP_STAMPS"; run with DASP_AMD_SO=dasp_amd/variants/stamps/libdasp_amd.so): every workgroup
of dasp_spmv_kernel / dasp_spmv_win1_kernel records the 100 MHz wall clock at entry and exit, the shader clock at entry, after the argument / table loads
(windowed: before the x copy), after the x copy, behind the barrier, at the end of wave 0 and of the first / last wave, its XCD and its CU.  The launches run
back to back as in the timing protocol; the table is over the last launches."""
import ctypes as C
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import dasp_amd as D


def pct(a, qs=(0, 10, 50, 90, 100)):
    a = np.asarray(a, np.float64)
    return " ".join("%7.2f" % np.percentile(a, q) for q in qs)


def main():
    name, prec = sys.argv[1], int(sys.argv[2])
    opts = {}
    scale = 1.0
    for kv in sys.argv[3:]:
        k, v = kv.split("=")
        if k == "scale":
            scale = float(v)
        else:
            opts[k] = int(v)
    L = D._lib.lib()
    if not hasattr(L, "dasp_debug_set_stamps"):
        sys.exit("not the DASP_STAMPS build: set DASP_AMD_SO=dasp_amd/variants/stamps/libdasp_amd.so")
    dt, tdt = (np.float64, torch.float64) if prec == 64 else (np.float16, torch.float16)
    rp, ci = D.synth_csr(name, scale)
    m, n = D.synth_dims(name, scale)[:2]
    plan = D.Plan(rp, ci, np.ones(ci.size, dt), n, precision=prec, **opts).upload()
    st = plan.stats
    x = torch.ones(n, dtype=tdt, device="cuda")
    y = torch.zeros(m, dtype=tdt, device="cuda")
    launches = 24
    ms_plain = plan.time(x.data_ptr(), y.data_ptr(), 0, 100, 1000)[1]       # the stamped build with the stamps off (compare with the product's number)
    grid_guess = st["n_workgroups"] + 64
    wpw = min(16, st["row_window"] // 16) if st["x_window_on"] else 4
    cap = launches * grid_guess * wpw
    rec = torch.zeros(cap * 12, dtype=torch.int64, device="cuda")
    assert L.dasp_debug_set_stamps(C.c_void_p(rec.data_ptr()), C.c_uint(cap)) == 0
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    ev0.record()
    for _ in range(launches):
        plan.spmv(x.data_ptr(), y.data_ptr(), 0)
    ev1.record()
    torch.cuda.synchronize()
    ms_stamped = ev0.elapsed_time(ev1) / launches
    assert L.dasp_debug_set_stamps(C.c_void_p(0), C.c_uint(0)) == 0
    r = rec.cpu().numpy().reshape(cap, 12).view(np.uint64)
    r = r[(r[:, 0] >> np.uint64(63)) == 1]
    grid = int(r[0, 2])
    launch = r[:, 10].astype(np.int64)
    nl = int(launch.max()) + 1
    blk = (r[:, 0] & np.uint64(0xFFFFFFFF)).astype(np.int64)
    wave = ((r[:, 0] >> np.uint64(32)) & np.uint64(0xFF)).astype(np.int64)
    kind = ((r[:, 0] >> np.uint64(48)) & np.uint64(0xFF)).astype(np.int64)
    xcc = (r[:, 1] & np.uint64(0xF)).astype(np.int64)
    hw = (r[:, 1] >> np.uint64(32)).astype(np.int64)
    wall0, wall1 = r[:, 3].astype(np.int64), r[:, 4].astype(np.int64)
    clk = r[:, 5:10].astype(np.int64)            # entry, tables there (windowed), x copy done, behind the barrier, end
    print("%s f%d scale %g %s: %d workgroups per launch, %d launches, %d wave records; stamped build: %.2f us per launch with the stamps off, %.2f us with them on"
          % (name, prec, scale, opts, grid, nl, len(r), 1e3 * ms_plain, 1e3 * ms_stamped))
    print("plan: windows %d row_window %d lds %d  blocks %d pieces %d short tiles %d" % (st["n_windows"], st["row_window"], st["lds_bytes"], st["n_med_blocks"], st["n_long_pieces"], st["n_short_tiles"]))
    rows = []
    for k in range(nl // 2, nl - 1):
        sel, nx = launch == k, launch == k + 1
        w0, w1 = wall0[sel], wall1[sel]
        rows.append(((w1.max() - w0.min()) / 100.0, (w0.max() - w0.min()) / 100.0, (w1.min() - w0.min()) / 100.0, (wall0[nx].min() - w1.max()) / 100.0,
                     (wall0[nx].min() - w0.min()) / 100.0))
    rows = np.array(rows)
    print("per launch (us; median over %d launches): first entry -> last exit %.2f | first -> last wave start %.2f | first wave exit at %.2f | last exit -> next launch's first entry %.2f | launch period %.2f"
          % (len(rows), np.median(rows[:, 0]), np.median(rows[:, 1]), np.median(rows[:, 2]), np.median(rows[:, 3]), np.median(rows[:, 4])))
    # shader clock from the longest waves
    d = wall1 - wall0
    i = np.argsort(d)[-200:]
    ghz = float(np.median((clk[i, 4] - clk[i, 0]) / (d[i] * 10.0)))
    print("shader clock ~ %.2f GHz" % ghz)
    us = lambda c: np.asarray(c, np.float64) / (ghz * 1e3)
    sel = launch == nl - 2
    t0 = wall0[sel].min()
    print("launch %d, percentiles 0 / 10 / 50 / 90 / 100 over its waves (us):" % (nl - 2))
    for kd, label in ((3, "window"), (1, "medium"), (2, "short"), (0, "long")):
        m = sel & (kind == kd)
        if not m.any():
            continue
        print(" %-8s waves=%d" % (label, int(m.sum())))
        print("   start after the launch's first wave    %s" % pct((wall0[m] - t0) / 100.0))
        if kd == 3:
            print("   entry -> arguments + window table there %s" % pct(us(clk[m, 1] - clk[m, 0])))
            print("   x copy (loads returned, LDS written)    %s" % pct(us(clk[m, 2] - clk[m, 1])))
            print("   wait at the barrier                     %s" % pct(us(clk[m, 3] - clk[m, 2])))
            print("   blocks (tiles -> LDS gathers -> MFMA -> y) %s" % pct(us(clk[m, 4] - clk[m, 3])))
            # per workgroup: spread of its waves' block phases
            key = blk[m]
            o = np.argsort(key, kind="stable")
            e = us(clk[m, 4] - clk[m, 3])[o]
            ks = key[o]
            cut = np.flatnonzero(np.diff(ks)) + 1
            grp = np.split(e, cut)
            print("   per workgroup: shortest wave's blocks   %s" % pct([g.min() for g in grp]))
            print("   per workgroup: longest wave's blocks    %s" % pct([g.max() for g in grp]))
        print("   whole wave                              %s" % pct(us(clk[m, 4] - clk[m, 0])))
        print("   exit after the launch's first entry     %s" % pct((wall1[m] - t0) / 100.0))
    cu = (xcc[sel] << 16) | (hw[sel] & 0xFF00)
    wg_of = blk[sel]
    pairs = np.unique(np.stack([cu, wg_of], 1), axis=0)
    _, cnt = np.unique(pairs[:, 0], return_counts=True)
    print("placement: %d distinct CUs used; workgroups per used CU: max %d, histogram %s" % (len(cnt), cnt.max(), np.bincount(cnt).tolist()))
    for xc in range(8):
        m = sel & (xcc == xc)
        if m.any():
            print("  XCD %d: %d waves, last exit %.2f us after the launch's first entry" % (xc, int(m.sum()), (wall1[m].max() - t0) / 100.0))
    plan.close()


if __name__ == "__main__":
    main()

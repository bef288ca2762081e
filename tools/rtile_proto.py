#!/usr/bin/env python3
"""tools/rtile_proto.py -- prototype: the column panels of an f16 plan with ONE shared row order.  A panel's rows of at
most T nonzeros are stored in the parent's slot order in tiles of 64 slots (tools/proto/rtile_proto.hip: one wave per tile, one complete line
of partial y per wave); the rows above T stay with a plan of the library's kernels.  Timed against the product's plan of the same matrix.
Needs dasp_amd/variants/proto/librtile.so (hipcc line in the .hip file)."""
import ctypes, os, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import dasp_amd as D
lib = ctypes.CDLL(os.path.join(ROOT, "dasp_amd/variants/proto/librtile.so"))
lib.rtile_launch.argtypes = [ctypes.c_int] + [ctypes.c_void_p] * 5 + [ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p]
lib.rtile_launch_b.argtypes = [ctypes.c_int, ctypes.c_int] + [ctypes.c_void_p] * 5 + [ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p]
lib.rtile_sum.argtypes = [ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int, ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p]

def gpu_time(fn, warm=5, iters=50):
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / iters

name = sys.argv[1] if len(sys.argv) > 1 else "ljournal-2008"
Ts = [int(t) for t in sys.argv[2].split(",")] if len(sys.argv) > 2 else [8, 12, 16]
rows, cols = D.synth_dims(name, 1.0)
rp, ci = D.synth_csr(name, 1.0)
rng = np.random.default_rng(3)
val = rng.uniform(0.5, 1.5, ci.size).astype(np.float16)
xh = rng.uniform(0.5, 1.5, cols).astype(np.float16)
parent = D.Plan(rp, ci, val, cols, precision=16)
P = parent.stats["n_col_panels"]
print(name, "rows", rows, "nnz", ci.size, "panels", P, flush=True)
if P < 2: sys.exit("the plan has no column panels")
slot_of_row = np.argsort(parent.order_rid).astype(np.int64)
parent.upload(); parent.drop_host()
x = torch.from_numpy(xh).cuda()
y = torch.zeros(rows, dtype=torch.float16, device="cuda")
t_prod = gpu_time(lambda: parent.spmv(x.data_ptr(), y.data_ptr()))
print("product plan: %.4f ms" % t_prod, flush=True)
lens = np.diff(rp)
want = np.add.reduceat(val.astype(np.float64) * xh[ci].astype(np.float64), np.minimum(rp[:-1], ci.size - 1)) * (lens > 0)
got = y.cpu().numpy().astype(np.float64)
print("product max rel err %.2e" % np.max(np.abs(got - want) / np.maximum(1, np.abs(want))), flush=True)

w = -(-cols // P)
rowid = np.repeat(np.arange(rows, dtype=np.int32), lens)
pan = (ci // w).astype(np.int64)
key = pan * rows + slot_of_row[rowid]                       # (panel, parent slot) of every entry
cnt = np.bincount(key, minlength=P * rows).astype(np.int32)        # nonzeros per (panel, slot)
t0 = time.time(); order = np.argsort(key, kind="stable"); print("sort %.1f s" % (time.time() - t0), flush=True)
n_tiles = -(-rows // 64)
slots_pad = n_tiles * 64
for T in Ts:
    keep_slot = cnt <= T
    keep = keep_slot[key]
    dev, rest = [], []
    part = torch.zeros(P * slots_pad, dtype=torch.float16, device="cuda")
    yrest = torch.zeros(rows, dtype=torch.float16, device="cuda")
    for k in range(P):
        kc = np.zeros(slots_pad, np.int64); kc[:rows] = np.where(keep_slot[k * rows:(k + 1) * rows], cnt[k * rows:(k + 1) * rows], 0)
        kc = kc.reshape(n_tiles, 64)
        tile_ptr = np.zeros(n_tiles + 1, np.int64); np.cumsum(kc.sum(1), out=tile_ptr[1:])
        rowstart = (np.cumsum(kc, 1) - kc).astype(np.uint16)
        mk = np.zeros(slots_pad, bool); mk[:rows] = keep_slot[k * rows:(k + 1) * rows]
        mask = (mk.reshape(n_tiles, 64).astype(np.uint64) << np.arange(64, dtype=np.uint64)).sum(1, dtype=np.uint64)
        sel = order[(key[order] >= k * rows) & (key[order] < (k + 1) * rows)]
        sel = sel[keep[sel]]
        assert sel.size == tile_ptr[-1]
        dev.append([torch.from_numpy(a).cuda() for a in (val[sel], ci[sel].astype(np.int32), rowstart.reshape(-1), mask.view(np.int64), tile_ptr.astype(np.int32))])
        r = (pan == k) & ~keep                                      # the rows above T: a plan of the library's kernels
        rl = np.bincount(rowid[r], minlength=rows)
        rrp = np.zeros(rows + 1, np.int64); np.cumsum(rl, out=rrp[1:])
        if r.any():
            q = D.Plan(rrp.astype(np.int32), ci[r], val[r], cols, precision=16, col_panels=1).upload(); q.drop_host()
            rest.append(q)
        print("  T %d panel %d: %d of %d nonzeros in tiles, %d rows above T" % (T, k, sel.size, int((pan == k).sum()), int((~keep_slot[k * rows:(k + 1) * rows]).sum())), flush=True)
    def tiles():
        for k in range(P):
            v, c, rs, m, tp = dev[k]
            lib.rtile_launch(T, v.data_ptr(), c.data_ptr(), rs.data_ptr(), m.data_ptr(), tp.data_ptr(), n_tiles, x.data_ptr(), part.data_ptr() + 2 * k * slots_pad, None)
    def tiles_b(R):
        def go():
            for k in range(P):
                v, c, rs, m, tp = dev[k]
                assert lib.rtile_launch_b(T, R, v.data_ptr(), c.data_ptr(), rs.data_ptr(), m.data_ptr(), tp.data_ptr(), n_tiles, x.data_ptr(), part.data_ptr() + 2 * k * slots_pad, None) == 0
        return go
    def rests():
        for q in rest: q.spmv(x.data_ptr(), yrest.data_ptr())
    def total():
        tiles(); rests(); lib.rtile_sum(part.data_ptr(), slots_pad, P, y.data_ptr(), slots_pad if slots_pad <= rows else rows // 8 * 8, None)
    s2 = torch.cuda.Stream(); cur = torch.cuda.current_stream()
    def total2():          # tiles on a second stream beside the rest plans: what one fused launch per panel would overlap
        s2.wait_stream(cur)
        for k in range(P):
            v, c, rs, m, tp = dev[k]
            lib.rtile_launch(T, v.data_ptr(), c.data_ptr(), rs.data_ptr(), m.data_ptr(), tp.data_ptr(), n_tiles, x.data_ptr(), part.data_ptr() + 2 * k * slots_pad, s2.cuda_stream)
        rests(); cur.wait_stream(s2)
        lib.rtile_sum(part.data_ptr(), slots_pad, P, y.data_ptr(), slots_pad if slots_pad <= rows else rows // 8 * 8, None)
    print("  tiles beside the rest plans (two streams) + sum: %.4f ms" % gpu_time(total2), flush=True)
    tt, tr, ta = gpu_time(tiles), gpu_time(rests), gpu_time(total)
    print("T %d: tiles %.4f ms, rows above T (library plans, %d launches) %.4f ms, tiles + rest + sum %.4f ms   (product %.4f)" % (T, tt, len(rest), tr, ta, t_prod), flush=True)
    if T <= 16:
        for R in (-2,):
            print("  variant %s, %d slots per wave: tiles %.4f ms" % ("B" if R > 0 else "C", 64 * abs(R), gpu_time(tiles_b(R))), flush=True)
            part.zero_(); tiles_b(R)(); torch.cuda.synchronize()
            ps = part.view(P, slots_pad)[:, :rows].float().sum(0).cpu().numpy().astype(np.float64)
            wk = np.bincount(slot_of_row[rowid[keep]], weights=(val.astype(np.float64) * xh[ci].astype(np.float64))[keep], minlength=rows)
            print("    max rel err %.2e" % np.max(np.abs(ps - wk) / np.maximum(1, np.abs(wk))), flush=True)
    # check the tile part: tiles + sum against the kept entries
    part.zero_(); tiles(); torch.cuda.synchronize()
    ps = part.view(P, slots_pad)[:, :rows].float().sum(0).cpu().numpy().astype(np.float64)
    wk = np.bincount(slot_of_row[rowid[keep]], weights=(val.astype(np.float64) * xh[ci].astype(np.float64))[keep], minlength=rows)
    print("  tile part max rel err %.2e" % np.max(np.abs(ps - wk) / np.maximum(1, np.abs(wk))), flush=True)
    for q in rest: q.close()
    del dev, rest, part

#!/usr/bin/env python3
"""tools/micro/pb_probe.py [workload=ljournal-2008] [scale=1.0] [B=8192] [C=4096]: propagation blocking (tools/micro/pb_kernels.hip) against the library's
f16 plan on the same stand-in: an EXPERIMENT on the gather-bound graph matrices, not part of the product.  Prints the time of both
phases, the bytes they move per nonzero, the library's time, and the worst error of both against a float64 CSR product on the f16-rounded inputs."""
import ctypes as C, os, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import dasp_amd as D
name = sys.argv[1] if len(sys.argv) > 1 else "ljournal-2008"
scale = float(sys.argv[2]) if len(sys.argv) > 2 else 1.0
B = int(sys.argv[3]) if len(sys.argv) > 3 else 8192
Cc = int(sys.argv[4]) if len(sys.argv) > 4 else 4096
lib = C.CDLL(os.path.join(ROOT, "build", "micro", "libpb.so"))
lib.pb_time.restype = C.c_float
m, n = D.synth_dims(name, scale)
rp, ci = D.synth_csr(name, scale)
nnz = int(rp[-1])
rng = np.random.default_rng(3)
val = (rng.integers(1, 9, nnz) / 64.0).astype(np.float16)
x = rng.uniform(0.5, 1.5, n).astype(np.float16)
t0 = time.time()
row = np.repeat(np.arange(m, dtype=np.int32), np.diff(rp))
nb, ncb = (m + B - 1) // B, (n + Cc - 1) // Cc
cb, bn = ci // Cc, row // B
key1 = cb.astype(np.int64) * nb + bn                     # phase-1 order: column block, then bin
o1 = np.argsort(key1, kind="stable")
k1 = key1[o1]
# tiles = runs of equal key in phase-1 order
starts = np.flatnonzero(np.r_[True, k1[1:] != k1[:-1]])
lens = np.diff(np.r_[starts, nnz]).astype(np.int32)
t_cb, t_bn = (k1[starts] // nb).astype(np.int64), (k1[starts] % nb).astype(np.int64)
# phase-2 order: bin, then column block: tiles sorted by (bin, cb); a tile's destination = prefix of the lengths in that order
o_t = np.lexsort((t_cb, t_bn))
dst_of_tile = np.empty(starts.size, np.int64)
dst_of_tile[o_t] = np.r_[0, np.cumsum(lens[o_t])[:-1]]
tiles = np.stack([starts.astype(np.int32), dst_of_tile.astype(np.int32), lens], axis=1).astype(np.int32).copy()
tile_ptr = np.searchsorted(t_cb, np.arange(ncb + 1)).astype(np.int32)
cb_ptr = np.r_[starts, nnz][tile_ptr].astype(np.int32)
dst = (np.repeat(dst_of_tile, lens) + (np.arange(nnz) - np.repeat(starts, lens))).astype(np.int32)      # per nonzero, phase-1 order
val1, lcol1 = val[o1], (ci[o1] % Cc).astype(np.uint16)
lrow2 = np.empty(nnz, np.uint16); lrow2[dst] = (row[o1] % B).astype(np.uint16)
bin_ptr = np.r_[0, np.cumsum(np.bincount(bn, minlength=nb))].astype(np.int32)
print("%s x%g: %d rows, %d nnz; B=%d (%d bins), C=%d (%d column blocks), %d tiles (mean %.0f nonzeros); layout built in %.1f s"
      % (name, scale, m, nnz, B, nb, Cc, ncb, starts.size, nnz / starts.size, time.time() - t0), flush=True)
dev = lambda a: torch.from_numpy(np.ascontiguousarray(a).view(np.int16 if a.dtype in (np.float16, np.uint16) else a.dtype)).cuda()
d = {k: dev(v) for k, v in dict(val=val1, lcol=lcol1, tiles=tiles, tile_ptr=tile_ptr, dst=dst, cb_ptr=cb_ptr, x=x, lrow=lrow2, bin_ptr=bin_ptr).items()}
contrib = torch.zeros(nnz, dtype=torch.float32, device="cuda"); y = torch.zeros(m, dtype=torch.float16, device="cuda")
p = lambda t: C.c_void_p(t.data_ptr())
# reference: float64 CSR product of the f16-rounded inputs, and sum |a x| per row
import scipy.sparse as sp
A = sp.csr_matrix((val.astype(np.float64), ci.copy(), rp.copy()), shape=(m, n))       # copies: abs() below sums duplicates IN PLACE
want = A @ x.astype(np.float64)
scl = np.maximum(abs(A) @ np.abs(x.astype(np.float64)), 1e-30)
for flat in (0, 1):
    ms1, ms2 = C.c_float(), C.c_float()
    ms = lib.pb_time(p(d["val"]), p(d["lcol"]), p(d["tiles"]), p(d["tile_ptr"]), p(d["dst"]), p(d["cb_ptr"]), p(d["x"]), Cc, n, ncb, p(contrib), p(d["lrow"]),
                     p(d["bin_ptr"]), B, m, nb, p(y), flat, 5, 50, C.byref(ms1), C.byref(ms2))
    got = y.cpu().numpy().astype(np.float64)
    bytes_nnz = (4 + 4 + 6) + (4 if flat else 0)
    print("propagation blocking, phase 1 %s: %.4f ms = %.4f + %.4f; %d B per nonzero -> %.2f TB/s; max error %.2e of sum|a x|"
          % ("flat (one thread per nonzero)" if flat else "tiled (a wave per tile)", ms, ms1.value, ms2.value, bytes_nnz, nnz * bytes_nnz / ms / 1e9,
             float((np.abs(got - want) / scl).max())), flush=True)
plan = D.Plan(rp, ci, val, n, precision=16, y_order=D.Y_NATURAL).upload()
xx = torch.from_numpy(x.view(np.int16)).cuda(); yy = torch.zeros(m + 64, dtype=torch.int16, device="cuda")
e = plan.time(xx.data_ptr(), yy.data_ptr(), 0, 5, 50)[1]
got = yy[:m].cpu().numpy().view(np.float16).astype(np.float64)
print("library (DASP f16 plan, %d column panels): %.4f ms; max error %.2e of sum|a x|" % (plan.stats.get("n_col_panels", 0), e, float((np.abs(got - want) / scl).max())))

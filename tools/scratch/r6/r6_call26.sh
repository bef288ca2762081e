#!/bin/bash
# r6_call26 -- experiment build (-DDASP_EXP_S16): the wave-segmented short tiles (f64, four tiles per wave) with 16-bit ids from a per-tile base, fabricated at upload;
# the short-rows-only families of tools/category_sweep.py, same library with the knob off / on
export PYTHONPATH=$PWD
cat > /tmp/s16.py <<'PY'
import os, sys, numpy as np, torch
import dasp_amd as D
class C:
    rng = np.random.default_rng(11)
    @staticmethod
    def from_lengths(lens, n, band):
        rng = C.rng
        m = lens.size
        rp = np.zeros(m + 1, np.int64); np.cumsum(lens, out=rp[1:])
        rows = np.repeat(np.arange(m, dtype=np.int64), lens)
        k = np.arange(int(rp[-1]), dtype=np.int64) - rp[rows]
        centre = rows * n // max(m, 1)
        start = np.clip(centre + rng.integers(-band, band + 1, m)[rows] - lens[rows] // 2, 0, np.maximum(n - lens[rows], 0))
        return rp.astype(np.int32), (start + k).astype(np.int32)
M = 1 << 20
fams = [("1..4", lambda: C.from_lengths(C.rng.integers(1, 5, 24 * M), 24 * M, 64), 24 * M), ("all 3", lambda: C.from_lengths(np.full(24 * M, 3), 24 * M, 64), 24 * M),
        ("all 1", lambda: C.from_lengths(np.full(48 * M, 1), 48 * M, 64), 48 * M), ("1..4 x0.1", lambda: C.from_lengths(C.rng.integers(1, 5, 2 * M), 2 * M, 64), 2 * M)]
for name, mk, n in fams:
    rp, ci = mk(); m = rp.size - 1
    p = D.Plan(rp, ci, np.ones(ci.size, np.float64), n, precision=64).upload()
    x = torch.ones(n, dtype=torch.float64, device="cuda"); y = torch.zeros(m, dtype=torch.float64, device="cuda")
    p.spmv(x.data_ptr(), y.data_ptr(), 0); torch.cuda.synchronize()
    yy = y.cpu().numpy(); want = np.diff(rp).astype(np.float64)
    perm = p.order_rid
    ok = np.array_equal(yy, want[perm])
    t = sorted(p.time(x.data_ptr(), y.data_ptr(), 0, 10, 50)[1] for _ in range(3))
    print(sys.argv[1], name, "seg", p.stats.get("short_seg"), "exact", ok, "%.1f us" % (t[0] * 1e3), flush=True)
    p.close()
PY
export DASP_AMD_SO=$PWD/dasp_amd/variants/s16/libdasp_amd.so
for r in 1 2; do
python3 /tmp/s16.py off 2>&1 | grep -v amdgpu.ids
DASP_EXP_S16=1 python3 /tmp/s16.py on 2>&1 | grep -v amdgpu.ids
done

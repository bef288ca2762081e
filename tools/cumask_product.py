#!/usr/bin/env python3
"""tools/cumask_product.py [workload=HV15R]: what the f64 product loses when its stream is kept off k CUs of every XCD (hipExtStreamCreateWithCUMask;
KFD deals mask bit i to XCD i % 8, then to the XCD's shader engines in turn: tools/micro/cumask.hip) -- the whole matrix and rank 3's share of an 8-way
partition (the multi-GPU step's product)."""
import ctypes, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import dasp_amd as D
hip = ctypes.CDLL("libamdhip64.so")
name = sys.argv[1] if len(sys.argv) > 1 else "HV15R"
cus = torch.cuda.get_device_properties(0).multi_processor_count


def masked_stream(per_xcd):
    m = (ctypes.c_uint32 * ((cus + 31) // 32))(*([0xFFFFFFFF] * ((cus + 31) // 32)))
    for b in range(cus - 8 * per_xcd, cus):
        m[b // 32] &= ~(1 << (b % 32))
    s = ctypes.c_void_p()
    rc = hip.hipExtStreamCreateWithCUMask(ctypes.byref(s), len(m), m)
    assert rc == 0, rc
    return s.value


streams = {k: masked_stream(k) for k in (0, 1, 2, 4, 8)}
rows, cols = D.synth_dims(name, 1.0)
lengths = D.synth_row_lengths(name, 1.0)
rpf = np.zeros(rows + 1, np.int64); np.cumsum(lengths, out=rpf[1:])
b = np.searchsorted(rpf, rpf[-1] * np.arange(9) // 8, side="left")
for tag, r0, r1 in (("whole", 0, rows), ("rank 3 of 8", int(b[3]), int(b[4]))):
    rp, ci = D.synth_csr(name, 1.0, r0, r1, lengths=lengths[r0:r1])
    p = D.Plan(rp, ci, np.ones(ci.size), cols).upload(); p.drop_host()
    x = torch.ones(cols, dtype=torch.float64, device="cuda"); y = torch.zeros(r1 - r0, dtype=torch.float64, device="cuda")
    for rnd in range(2):
        line = "%s %-12s round %d:" % (name, tag, rnd)
        for k, s in streams.items():
            e = p.time(x.data_ptr(), y.data_ptr(), s, 20, 200)[1]
            line += "  %3d CUs %.4f ms" % (cus - 8 * k, e)
        print(line, flush=True)
    p.close(); del x, y, rp, ci

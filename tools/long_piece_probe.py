#!/usr/bin/env python3
"""tools/long_piece_probe.py <workload> <prec> <long_piece> -- 300 SpMVs of one plan (for rocprofv3 --kernel-trace --stats: what the main kernel alone takes
when the long rows are cut into pieces of long_piece, next to the stage-2 launch it then needs)."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import dasp_amd as D
name, prec, piece = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
rows, cols = D.synth_dims(name, 1.0)
rp, ci = D.synth_csr(name, 1.0)
dt, tdt = (np.float64, torch.float64) if prec == 64 else (np.float16, torch.float16)
p = D.Plan(rp, ci, np.ones(ci.size, dt), cols, precision=prec, long_piece=piece)
st = p.stats
p.upload()
x = torch.ones(cols, dtype=tdt, device="cuda"); y = torch.zeros(rows, dtype=tdt, device="cuda")
print(name, prec, "long_piece", piece, "pieces", st["n_long_pieces"], "multi", st["n_long_multi"], "ms %.4f" % p.time(x.data_ptr(), y.data_ptr(), 0, 20, 300)[1], flush=True)

#!/bin/bash
# r6_call25 -- MedAhead (a wave's first block opened in front of the window's x copy, one-workgroup-per-CU build): correctness, then A/B against HEAD's library on one box
export PYTHONPATH=$PWD
out=gpurun_out/r6; mkdir -p $out
timeout 900 python3 -m pytest tests/test_gpu_spmv.py -m gpu -x -q -k "window or win or hybrid or golden or bit_identical" 2>&1 | tail -4
cat > /tmp/ab25.py <<'PY'
import sys, numpy as np, torch, dasp_amd as D
tag = sys.argv[1]
for name, prec, sc in (("cop20k_A",64,1.0),("cop20k_A",64,2.0),("cop20k_A",64,0.5),("cop20k_A",16,1.0),("cop20k_A",16,4.0),("HV15R",64,0.01),("cop20k_A",64,4.0)):
    dt, tdt = (np.float64, torch.float64) if prec == 64 else (np.float16, torch.float16)
    rp, ci = D.synth_csr(name, sc); m, n = D.synth_dims(name, sc)[:2]
    p = D.Plan(rp, ci, np.ones(ci.size, dt), n, precision=prec).upload()
    x = torch.ones(n, dtype=tdt, device="cuda"); y = torch.zeros(m, dtype=tdt, device="cuda")
    t = sorted(p.time(x.data_ptr(), y.data_ptr(), 0, 50, 500)[1] for _ in range(5))
    print(tag, name, "f%d" % prec, sc, "windows", p.stats["x_window_on"], p.stats["n_windows"], "%.2f us (median %.2f)" % (t[0] * 1e3, t[2] * 1e3), flush=True)
    p.close()
PY
for r in 1 2; do
  DASP_AMD_SO=$PWD/dasp_amd/variants/r6base/libdasp_amd.so python3 /tmp/ab25.py base 2>&1 | grep -v amdgpu.ids
  python3 /tmp/ab25.py ahead 2>&1 | grep -v amdgpu.ids
done

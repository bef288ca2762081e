#!/bin/bash
# r6_call21 -- the core-span condition of the window rule: cop20k_A in f16 (core span 12.8 KB) with and without windows, three sizes; the band + outliers family; the guard
export PYTHONPATH=$PWD
out=gpurun_out/r6
python3 - <<'PY' 2>&1 | grep -v amdgpu.ids
import numpy as np, torch, dasp_amd as D
for name, prec, sc in (("cop20k_A",16,1.0),("cop20k_A",16,4.0),("cop20k_A",16,16.0),("cop20k_A",64,1.0)):
    dt, tdt = (np.float64, torch.float64) if prec == 64 else (np.float16, torch.float16)
    rp, ci = D.synth_csr(name, sc); m, n = D.synth_dims(name, sc)[:2]
    for kw in ({}, {"x_window": 81920}, {"x_window": -1}):
        p = D.Plan(rp, ci, np.ones(ci.size, dt), n, precision=prec, **kw).upload()
        x = torch.ones(n, dtype=tdt, device="cuda"); y = torch.zeros(m, dtype=tdt, device="cuda")
        t = min(p.time(x.data_ptr(), y.data_ptr(), 0, 50, 500)[1] for _ in range(3)) * 1e3
        print(name, "f%d" % prec, sc, kw, "windows", p.stats["x_window_on"], p.stats["n_windows"], "%.2f us" % t, flush=True)
        p.close()
PY
mkdir -p $out; bash tools/scratch/r6/r6_call19.sh
timeout 1500 python3 -m pytest tests/test_zz_auto_rules.py -m gpu -x -q > $out/autorules21.log 2>&1; tail -4 $out/autorules21.log
cp gpurun_out/r6_auto_rules.md $out/auto_rules21.md; grep LOSS $out/auto_rules21.md

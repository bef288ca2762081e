#!/bin/bash
# r6_call6 -- one gpurun batch of round 6 (its output: gpurun_out/r6/; what it measured is quoted in profiles/r06_*.md)
out=gpurun_out/r6; mkdir -p $out
export PYTHONPATH=$PWD
cat > /tmp/r6_rot.py <<'PY'
import sys, os, numpy as np, torch, dasp_amd as D
tag = os.environ.get("DASP_WG_ROT", "0")
for name, prec, sc in (("webbase-1M",16,1.0),("webbase-1M",64,1.0),("webbase-1M-uniform",16,1.0),("powerlaw_1M",64,0.03),("HV15R",64,0.1),("nlpkkt160",64,0.03),("webbase-1M",16,4.0)):
    dt, tdt = (np.float64, torch.float64) if prec == 64 else (np.float16, torch.float16)
    rp, ci = D.synth_csr(name, sc); m, n = D.synth_dims(name, sc)[:2]
    p = D.Plan(rp, ci, np.ones(ci.size, dt), n, precision=prec, two_phase=-1).upload()
    x = torch.ones(n, dtype=tdt, device="cuda"); y = torch.zeros(m, dtype=tdt, device="cuda")
    t = [1e3 * p.time(x.data_ptr(), y.data_ptr(), 0, 100, 1000)[1] for _ in range(3)]
    lens = np.diff(rp)[p.order_rid].astype(np.float64)
    got = y.double().cpu().numpy()
    ok = bool(np.all((np.abs(got - lens) <= (0 if prec == 64 else 1e-2) * np.maximum(lens, 1)) | (lens > 2048)))
    print("rot", tag, name, prec, sc, "us", ["%.2f" % v for v in t], "ok" if ok else "WRONG", flush=True)
    p.close()
PY
for r in 0 1 2 0 1 2; do DASP_WG_ROT=$r timeout 300 python3 /tmp/r6_rot.py; done > $out/rot6.log 2>&1
grep -v amdgpu.ids $out/rot6.log
timeout 1500 python3 -m pytest tests/test_zz_auto_rules.py -m gpu -x -q > $out/autorules6.log 2>&1; tail -40 $out/autorules6.log
cp gpurun_out/r6_auto_rules.md $out/auto_rules6.md 2>/dev/null

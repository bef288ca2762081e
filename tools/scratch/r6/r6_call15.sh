#!/bin/bash
# r6_call15 -- per-block tail-step table (med_nt: scalar instead of a vector load in front of every block's stream): working tree against the commit before it
out=gpurun_out/r6; mkdir -p $out
export PYTHONPATH=$PWD
V=$PWD/dasp_amd/variants
cat > /tmp/r6_nt.py <<'PY'
import sys, os, numpy as np, torch, dasp_amd as D
tag = sys.argv[1]
for name, prec, sc in (("cop20k_A",64,1.0),("cop20k_A",64,4.0),("webbase-1M",16,1.0),("webbase-1M",64,1.0),("HV15R",64,0.01),("HV15R",64,0.1),("nlpkkt160",64,0.03),("powerlaw_1M",64,0.1),("rmat_2M",16,0.25),("HV15R",64,1.0),("nlpkkt160",64,1.0),("Queen_4147",64,1.0)):
    dt, tdt = (np.float64, torch.float64) if prec == 64 else (np.float16, torch.float16)
    rp, ci = D.synth_csr(name, sc); m, n = D.synth_dims(name, sc)[:2]
    p = D.Plan(rp, ci, np.ones(ci.size, dt), n, precision=prec, two_phase=-1).upload()
    p.drop_host()
    x = torch.ones(n, dtype=tdt, device="cuda"); y = torch.zeros(m, dtype=tdt, device="cuda")
    it = 1000 if ci.size < 2e7 else 100
    t = [1e3 * p.time(x.data_ptr(), y.data_ptr(), 0, it // 10, it)[1] for _ in range(3)]
    lens = torch.from_numpy(np.diff(rp)[p.order_rid].astype(np.float64)).cuda(); got = y.double()
    ok = bool((((got - lens).abs() <= (0 if prec == 64 else 1e-2) * lens.clamp(min=1)) | (lens > 2048)).all().item())
    print(tag, name, prec, sc, "us", ["%.2f" % v for v in t], "ok" if ok else "WRONG", flush=True)
    p.close(); del x, y; torch.cuda.empty_cache()
PY
for r in new prev new prev; do
  unset DASP_AMD_SO
  [ $r = prev ] && export DASP_AMD_SO=$V/prev/libdasp_amd.so
  timeout 900 python3 /tmp/r6_nt.py $r
done > $out/mednt15.log 2>&1
grep -v amdgpu.ids $out/mednt15.log | sort -k2,4 -s

#!/bin/bash
# r6_call13 -- one gpurun batch of round 6 (its output: gpurun_out/r6/; what it measured is quoted in profiles/r06_*.md)
out=gpurun_out/r6; mkdir -p $out
export PYTHONPATH=$PWD
V=$PWD/dasp_amd/variants
cat > /tmp/r6_m16.py <<'PY'
import sys, os, numpy as np, torch, dasp_amd as D
tag = sys.argv[1]
for name, sc in (("webbase-1M",1.0),("webbase-1M-uniform",1.0),("webbase-1M",4.0),("powerlaw_1M",0.1),("rmat_2M",0.25),("nlpkkt160",0.1),("HV15R",0.1)):
    rp, ci = D.synth_csr(name, sc); m, n = D.synth_dims(name, sc)[:2]
    p = D.Plan(rp, ci, np.ones(ci.size, np.float16), n, precision=16, two_phase=-1, col_panels=1).upload()
    x = torch.ones(n, dtype=torch.float16, device="cuda"); y = torch.zeros(m, dtype=torch.float16, device="cuda")
    it = 1000 if ci.size < 2e7 else 200
    t = [1e3 * p.time(x.data_ptr(), y.data_ptr(), 0, it // 10, it)[1] for _ in range(3)]
    lens = np.diff(rp)[p.order_rid].astype(np.float64); got = y.double().cpu().numpy()
    ok = bool(np.all((np.abs(got - lens) <= 1e-2 * np.maximum(lens, 1)) | (lens > 2048)))
    print(tag, name, sc, "us", ["%.2f" % v for v in t], "ok" if ok else "WRONG", flush=True)
    p.close()
PY
for r in multi16 single r5 multi16 single r5; do
  unset DASP_AMD_SO DASP_NO_SHORT_MULTI16
  [ $r = single ] && export DASP_NO_SHORT_MULTI16=1
  [ $r = r5 ] && export DASP_AMD_SO=$V/r5/libdasp_amd.so
  timeout 600 python3 /tmp/r6_m16.py $r
done > $out/multi16_13.log 2>&1
grep -v amdgpu.ids $out/multi16_13.log | sort -k2,3 -s

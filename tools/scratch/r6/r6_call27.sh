#!/bin/bash
# r6_call27 -- experiment builds (-DDASP_EXP_WPW=8 / 16): plain plans in workgroups of 8 / 16 waves instead of 4 -- is the launch of a small matrix bound by the
# dispatcher's WORKGROUP rate (webbase-1M f16: 3869 workgroups of 4 waves start over 10 us, profiles/r06_small_matrix.md section 5)?
export PYTHONPATH=$PWD
cat > /tmp/wpw.py <<'PY'
import sys, numpy as np, torch, dasp_amd as D
tag = sys.argv[1]
for name, prec, sc in (("webbase-1M",16,1.0),("webbase-1M",64,1.0),("webbase-1M",16,4.0),("HV15R",64,0.01),("HV15R",64,0.1),("powerlaw_1M",64,0.03),("rmat_2M",16,0.05),("cop20k_A",16,1.0),("HV15R",64,1.0),("nlpkkt160",16,1.0)):
    dt, tdt = (np.float64, torch.float64) if prec == 64 else (np.float16, torch.float16)
    rp, ci = D.synth_csr(name, sc); m, n = D.synth_dims(name, sc)[:2]
    p = D.Plan(rp, ci, np.ones(ci.size, dt), n, precision=prec).upload()
    x = torch.ones(n, dtype=tdt, device="cuda"); y = torch.zeros(m, dtype=tdt, device="cuda")
    p.spmv(x.data_ptr(), y.data_ptr(), 0); torch.cuda.synchronize()
    want = np.diff(rp).astype(np.float64)[p.order_rid]
    err = np.abs(y.double().cpu().numpy() - want).max() / max(want.max(), 1)
    it = (50, 500) if ci.size < 5e7 else (5, 30)
    t = sorted(p.time(x.data_ptr(), y.data_ptr(), 0, *it)[1] for _ in range(5))
    st = p.stats
    print(tag, name, "f%d" % prec, sc, "win", st["x_window_on"], "panels", st.get("col_panels"), "tp", st.get("two_phase"), "wgs", st["n_workgroups"], "err %.1e" % err, "%.2f us" % (t[0] * 1e3), flush=True)
    p.close()
PY
for r in 1 2; do
  DASP_AMD_SO=$PWD/dasp_amd/variants/wpw2/libdasp_amd.so python3 /tmp/wpw.py wpw2 2>&1 | grep -v amdgpu.ids
  python3 /tmp/wpw.py wpw4 2>&1 | grep -v amdgpu.ids
done

#!/bin/bash
# tools/r6_call5.sh -- GPU parity with the f16 hybrid; the hybrid against the pure two-phase / panel forms on the graph stand-ins; final stamps
out=gpurun_out/r6; mkdir -p $out
export PYTHONPATH=$PWD
V=$PWD/dasp_amd/variants
timeout 1200 python3 -m pytest tests -m gpu -x -q > $out/gputest5.log 2>&1; tail -3 $out/gputest5.log
cat > /tmp/r6_hyb.py <<'PY'
import sys, os, numpy as np, torch, dasp_amd as D
for name, sc, kws in (("rmat_2M", 1.0, ({}, {"long_cb": -1}, {"long_cb": 1})), ("powerlaw_1M", 1.0, ({}, {"two_phase": 1}, {"two_phase": 1, "long_cb": -1}, {"two_phase": -1})),
                      ("ljournal-2008", 1.0, ({}, {"long_cb": -1})), ("webbase-1M", 4.0, ({}, {"long_cb": -1})), ("rmat_2M", 0.5, ({}, {"long_cb": -1}))):
    rp, ci = D.synth_csr(name, sc); m, n = D.synth_dims(name, sc)[:2]
    for kw in kws:
        p = D.Plan(rp, ci, np.ones(ci.size, np.float16), n, precision=16, **kw).upload()
        x = torch.ones(n, dtype=torch.float16, device="cuda"); y = torch.zeros(m, dtype=torch.float16, device="cuda")
        t = [1e3 * p.time(x.data_ptr(), y.data_ptr(), 0, 20, 300)[1] for _ in range(3)]
        st = p.stats
        lens = np.diff(rp)[p.order_rid].astype(np.float64)
        got = y.double().cpu().numpy()
        ok = bool(np.all(np.abs(got - lens) <= 1e-2 * np.maximum(lens, 1)))
        b_alg = ci.size * 6 + (m + 1) * 4 + (n + m) * 2
        print(name, sc, kw, "two_phase", st["two_phase"], "panels", st["n_col_panels"], "lcb rows", st["lcb_rows"], "lcb elems", st["lcb_elems"], "us", ["%.1f" % v for v in t],
              "frac %.3f" % (b_alg / (min(t) * 1e-6) / 8e12), "ok" if ok else "WRONG", flush=True)
        p.close()
PY
timeout 1500 python3 /tmp/r6_hyb.py > $out/hybrid5.log 2>&1
grep -v amdgpu.ids $out/hybrid5.log
{
export DASP_AMD_SO=$V/stamps/libdasp_amd.so
echo "=== cop20k_A f64 auto (final r6 kernels)"; timeout 300 python3 tools/stamp_probe.py cop20k_A 64
echo "=== webbase-1M f16"; timeout 300 python3 tools/stamp_probe.py webbase-1M 16
} > $out/stamps5.log 2>&1
unset DASP_AMD_SO

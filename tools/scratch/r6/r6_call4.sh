#!/bin/bash
# tools/r6_call4.sh -- r6 window kernels: GPU parity, the 64-register build's batch / shot, tiles kept in the L2 (keep4 / keep2), the hub rows' side stream, L2 counters
out=gpurun_out/r6; mkdir -p $out
export PYTHONPATH=$PWD
V=$PWD/dasp_amd/variants
timeout 900 python3 -m pytest tests -m gpu -x -q > $out/gputest4.log 2>&1; tail -3 $out/gputest4.log
cat > /tmp/r6_time.py <<'PY'
import sys, os, numpy as np, torch, dasp_amd as D
tag = sys.argv[1]
cases = [("cop20k_A",64,1.0),("cop20k_A",64,2.0),("cop20k_A",64,4.0),("cop20k_A",64,16.0),("cop20k_A",64,64.0),("cop20k_A",16,1.0),("cop20k_A",16,4.0),("cop20k_A",16,16.0)]
if tag.startswith("keep"): cases = cases[:3]
for name, prec, sc in cases:
    dt, tdt = (np.float64, torch.float64) if prec == 64 else (np.float16, torch.float16)
    rp, ci = D.synth_csr(name, sc); m, n = D.synth_dims(name, sc)[:2]
    p = D.Plan(rp, ci, np.ones(ci.size, dt), n, precision=prec).upload()
    x = torch.ones(n, dtype=tdt, device="cuda"); y = torch.zeros(m, dtype=tdt, device="cuda")
    t = [1e3 * p.time(x.data_ptr(), y.data_ptr(), 0, 100, 1000)[1] for _ in range(3)]
    want = torch.from_numpy(np.diff(rp).astype(np.float64)[p.order_rid]).cuda()
    ok = bool((y.double() == want).all().item())
    print(tag, name, prec, sc, "windows", p.stats["n_windows"], "R", p.stats["row_window"], "us", ["%.2f" % v for v in t], "exact" if ok else "WRONG", flush=True)
    p.close()
PY
for v in product r5 w64_28 w64_36 keep4 keep2 product; do
  if [ $v = product ]; then unset DASP_AMD_SO; else export DASP_AMD_SO=$V/$v/libdasp_amd.so; fi
  timeout 900 python3 /tmp/r6_time.py $v
done > $out/variants4.log 2>&1
unset DASP_AMD_SO
grep -v amdgpu.ids $out/variants4.log
# ---- hub rows beside the panels (powerlaw_1M): side stream on / off, same box; y bit-equal between the two and from run to run
cat > /tmp/r6_pl.py <<'PY'
import sys, os, numpy as np, torch, dasp_amd as D
ys = {}
for name, prec, sc in (("powerlaw_1M",64,1.0),("powerlaw_1M",64,0.3),("powerlaw_1M",16,1.0)):
    dt, tdt = (np.float64, torch.float64) if prec == 64 else (np.float16, torch.float16)
    rp, ci = D.synth_csr(name, sc); m, n = D.synth_dims(name, sc)[:2]
    rng = np.random.default_rng(5)
    val = rng.uniform(0.5, 1.5, ci.size).astype(dt); xh = rng.uniform(0.5, 1.5, n).astype(dt)
    kw = {"two_phase": -1} if prec == 16 else {}
    p = D.Plan(rp, ci, val, n, precision=prec, **kw).upload()
    x = torch.from_numpy(xh).cuda(); y = torch.zeros(m, dtype=tdt, device="cuda")
    t = [1e3 * p.time(x.data_ptr(), y.data_ptr(), 0, 20, 200)[1] for _ in range(3)]
    g = 1e3 * p.time_graph(x.data_ptr(), y.data_ptr(), 0, 20, 200, 20)[1]
    y1 = y.clone(); y.zero_(); p.spmv(x.data_ptr(), y.data_ptr(), 0); torch.cuda.synchronize()
    print(os.environ.get("DASP_LCB_SIDE_STREAM", "1"), name, prec, sc, "panels", p.stats["n_col_panels"], "lcb rows", p.stats["lcb_rows"], "us", ["%.1f" % v for v in t], "graph %.1f" % g,
          "run-to-run bit-equal" if torch.equal(y, y1) else "RUNS DIFFER", "y hash", hash(y.cpu().numpy().tobytes()) & 0xffffffff, flush=True)
    p.close()
PY
for s in 1 0 1 0; do DASP_LCB_SIDE_STREAM=$s timeout 600 python3 /tmp/r6_pl.py; done > $out/sidestream4.log 2>&1
grep -v amdgpu.ids $out/sidestream4.log
# ---- L2 counters: the product against keep4 (every window but each 4th reads its tiles with plain loads)
export PMC_GROUPS="TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum;TCP_TCC_READ_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_PENDING_STALL_CYCLES_sum"
bash tools/pmc.sh r6_cop_product -- dasp_amd/bin/dasp_bench cop20k_A 1 64 200 20 > /dev/null 2>&1
LD_PRELOAD=$V/keep4/libdasp_amd.so bash tools/pmc.sh r6_cop_keep4 -- dasp_amd/bin/dasp_bench cop20k_A 1 64 200 20 > /dev/null 2>&1
LD_PRELOAD=$V/r5/libdasp_amd.so bash tools/pmc.sh r6_cop_r5 -- dasp_amd/bin/dasp_bench cop20k_A 1 64 200 20 > /dev/null 2>&1
for t in product keep4 r5; do echo "== $t"; cat gpurun_out/pmc_r6_cop_$t.txt; done

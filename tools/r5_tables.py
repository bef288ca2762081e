#!/usr/bin/env python3
"""tools/r5_tables.py -- the measured table of README.md / DESIGN.md from profiles/r05_bench_full.json.log (the full record of the round's final bench run); prints markdown"""
import json, os
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
D = json.load(open(os.path.join(root, "profiles", "r05_bench_full.json.log")))
r4 = {"HV15R": (0.4278, 0.976), "cop20k_A": (0.0107, 0.415), "nlpkkt160": (0.4290, 0.858), "powerlaw_1M": (0.5422, 0.268), "Queen_4147": (0.5080, 0.995), "HV15R-unstructured": (0.5066, 0.849),
      "webbase-1M": (0.0146, 0.253), "ljournal-2008": (0.4454, 0.153), "rmat_2M": (0.1347, 0.202), "ljournal-2008-uniform": (0.5046, 0.135), "webbase-1M-uniform": (0.0157, 0.235)}
form = {"cop20k_A": "LDS-staged x windows", "powerlaw_1M": "3 column panels in one launch + **column-blocked long rows**"}
rows = [("HV15R", "f64", D["roofline"]["kernel_ms"], D["roofline"]["frac"], D["roofline"].get("frac_random_values"), D["roofline"].get("frac_single_y"), D["roofline"].get("traffic_over_algorithmic"), 0)]
for e in D["suite"]:
    rows.append((e["workload"], e["dtype"], e["event_ms"], e["frac_hbm_roofline"], e.get("frac_hbm_roofline_random_values"), e.get("frac_single_y"), e.get("traffic_over_algorithmic"), e.get("two_phase")))
print("| stand-in | dtype | form (if not the plain DASP kernel) | r5 ms | r5 fraction of 8 TB/s | random values | one y (no choice among candidates) | counter traffic / B_alg | r4 ms / fraction |")
print("|---|---|---|---|---|---|---|---|---|")
for w, dt, ms, fr, rv, sy, tr, tp in rows:
    a, b = r4[w]
    print("| %s | %s | %s | %.4f | **%.3f** | %s | %s | %s | %.4f / %.3f |" % ("**HV15R** (bench headline)" if w == "HV15R" else w, dt, "**two-phase**" if tp else form.get(w, ""), ms, fr,
          "%.3f" % rv if rv else "--", "%.3f" % sy if sy else "--", "%.2f" % tr if tr else "--", a, b))
print("\ncpu_baseline %.2f GFLOP/s; rocSPARSE %s; f64 share >= 0.6: %s" % (D["cpu_baseline"]["value"], D.get("rocsparse_csr"), D["roofline"].get("f64_share_at_or_above_0.6")))

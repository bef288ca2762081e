#!/usr/bin/env python3
"""tools/readme_table.py: rewrite the r3 (a) / (b) cells of README.md's measured table from profiles/r03_bench_full_slowbox.json.log / r03_bench_full.json.log"""
import json, os, re
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
a = json.loads(open(os.path.join(root, "profiles/r03_bench_full_slowbox.json.log")).read())
b = json.loads(open(os.path.join(root, "profiles/r03_bench_full.json.log")).read())


def suite(d):
    return {s["workload"]: (s["event_ms"], s["frac_hbm_roofline"], s.get("frac_hbm_roofline_random_values", 0)) for s in d["suite"]}


def head(d, bold):
    r, rv, v = d["roofline"], d["roofline_random_values"], d["rocsparse_csr"]
    if bold:
        return ("**%.4f ms / %.3f** (mean of 200 back-to-back launches; single launches %.4f / %.4f / %.4f ms min / p90 / max); random values %.4f ms / %.3f; "
                "rocSPARSE CSR %.3f ms (%.2fx)" % (r["kernel_ms"], r["frac"], r["launch_ms_min"], r["launch_ms_p90"], r["launch_ms_max"], rv["kernel_ms"], rv["frac"],
                                                  v["ms"], v["speedup_of_dasp"]))
    return "%.4f / %.3f (random %.3f); rocSPARSE %.3f ms (%.2fx)" % (r["kernel_ms"], r["frac"], rv["frac"], v["ms"], v["speedup_of_dasp"])


sa, sb = suite(a), suite(b)
p = os.path.join(root, "README.md")
out = []
for l in open(p).read().split("\n"):
    m = re.match(r"\| (\*\*HV15R\*\* \(bench headline\)|[A-Za-z0-9_\-]+)( \([^|]*\))? \| (f64|f16) \| ([^|]*) \|", l)
    if m and l.count("|") >= 6:
        cells = [c.strip() for c in l.strip().strip("|").split("|")]
        if m.group(1).startswith("**HV15R**"):
            cells[3], cells[4] = head(a, True), head(b, False)
        elif m.group(1) in sa:
            cells[3], cells[4] = "%.4f / %.3f (%.3f)" % sa[m.group(1)], "%.4f / %.3f (%.3f)" % sb[m.group(1)]
        l = "| " + " | ".join(cells) + " |"
    out.append(l)
open(p, "w").write("\n".join(out))

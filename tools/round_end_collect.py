#!/usr/bin/env python3
"""tools/round_end_collect.py -- after `gpurun -- bash tools/round_end.sh`: turn gpurun_out/round_end_* into the committed profile files
(profiles/traffic.json + r03_traffic.md via traffic_collect.py, r03_hv15r_f64.md, r03_bench_full*.json.log, the all-ranks table of r03_multi_gpu_step.md)."""
import csv, glob, json, os, re, subprocess, sys
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
os.chdir(root)
if os.path.exists("gpurun_out/round_end_traffic.json"):       # generated on the GPU box, before the bench ran
    import shutil
    shutil.copy("gpurun_out/round_end_traffic.json", "profiles/traffic.json")
    shutil.copy("gpurun_out/round_end_traffic.md", "profiles/r03_traffic.md")
else:
    subprocess.check_call([sys.executable, "tools/traffic_collect.py"], stdout=subprocess.DEVNULL)
rows = list(csv.DictReader(open(glob.glob("gpurun_out/round_end_prof/*/*kernel_stats.csv")[0])))
line = [x for x in open("gpurun_out/round_end_prof.log") if x.startswith('{"metric')][-1].strip()
d = json.loads(line)
hv = [e for e in json.load(open("profiles/traffic.json")) if e["workload"] == "HV15R"][0]
r = d["roofline"]
out = ["# Round 3 profile of the bench command -- HV15R stand-in (2 017 169 rows, 275 454 726 nnz, f64), MI355X\n",
       "Source: `cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -- python3 bench.py --no-suite --no-vendor --steps 20` (kernel trace + stats only; the\n"
       "`--pmc FETCH_SIZE` / `--pmc WRITE_SIZE` passes are separate runs of `dasp_bench HV15R 1 64 20 3`, `profiles/r03_traffic.md`), kernel sources at `kernel_rev %s`\n"
       "(= `profiles/traffic.json`); all of it one batch on one box (`tools/round_end.sh`, `tools/round_end_collect.py`).  The %s profiled launches include the 200 back-to-back\n"
       "launches behind `roofline.kernel_ms`, the 200 timed one by one for the spread (`launch_ms_*`), those of the random-values plan and those of the placement trials (other allocations,\n"
       "some of them slower: `profiles/r03_placement.md`); the profiled average below (%.1f us over\n"
       "all of them) and the bench line's own `kernel_ms` of the same process (%.1f us; single launches min %.1f / p10 %.1f / median %.1f / p90 %.1f / max %.1f, each including the\n"
       "event between two kernels) agree.\n" % (hv["kernel_rev"], rows[0]["Calls"], float(rows[0]["AverageNs"]) / 1e3, r["kernel_ms"] * 1e3, r["launch_ms_min"] * 1e3,
                                            r["launch_ms_p10"] * 1e3, r["launch_ms_median"] * 1e3, r["launch_ms_p90"] * 1e3, r["launch_ms_max"] * 1e3),
       "## rocprofv3 --kernel-trace --stats --output-format csv -- python3 bench.py --no-suite --no-vendor --steps 20\n",
       "| kernel | calls | avg ns | min ns | max ns | % |\n|---|---|---|---|---|---|"]
for q in rows[:5]:
    out.append("| %s | %s | %.0f | %s | %s | %s |" % (q["Name"][:80], q["Calls"], float(q["AverageNs"]), q["MinNs"], q["MaxNs"], q["Percentage"]))
avg = float(rows[0]["AverageNs"])
out.append("\nRecomputed roofline fraction from the profile: %d B algorithmic / %.0f ns / 8 TB/s = **%.3f** (bench line: %.4f); counter traffic per launch %.4f GB = %.3f x algorithmic\n"
           "(`profiles/r03_traffic.md`: FETCH_SIZE %.4f GB raw x 2 + WRITE_SIZE %.2f MB), i.e. %.2f TB/s of real traffic.\n" %
           (r["algorithmic_bytes_per_launch"], avg, r["algorithmic_bytes_per_launch"] / avg / 8000, r["frac"], hv["traffic_bytes"] / 1e9,
            hv["traffic_bytes"] / r["algorithmic_bytes_per_launch"], hv["fetch_size_bytes_raw"] / 1e9, hv["write_size_bytes"] / 1e6, hv["traffic_bytes"] / avg / 1e3))
out.append("## bench.py's JSON line of the profiled run\n\n```\n%s\n```" % line)
open("profiles/r03_hv15r_f64.md", "w").write("\n".join(out) + "\n")
full = [x for x in open("gpurun_out/round_end_bench.json.log") if x.startswith("{")][-1]
D = json.loads(full)
slow = D["roofline"]["frac"] < 0.94                      # the two box populations: ~0.90-0.91 and ~0.97-0.98 on the HV15R headline
open("profiles/r03_bench_full_slowbox.json.log" if slow else "profiles/r03_bench_full.json.log", "w").write(full)
print("bench record -> %s box: %.4f ms, frac %.4f, traffic/alg %s" % ("slow" if slow else "fast", D["ms_per_step"], D["roofline"]["frac"], D["roofline"].get("traffic_over_algorithmic")))
for s in D["suite"]:
    print("  %-22s %.4f ms frac %.4f rand %.4f pre %.0f devpre %s traffic %s" % (s["workload"], s["event_ms"], s["frac_hbm_roofline"], s.get("frac_hbm_roofline_random_values", 0),
          s["pre_ms"], s.get("pre_ms_device_csr"), s.get("traffic_over_algorithmic")))
# all-ranks table
lines = [l.strip() for l in open("gpurun_out/round_end_mg_allranks.log") if l.startswith(("rank", "max over", "1-GPU"))]
tab = ["| rank | rows | nnz own / other columns | fused step, exchange 0 / 20 / 40 / 60 us | two launches + events, 0 / 20 / 40 / 60 us | step kernel alone | own-column alone | other-column alone |",
       "|---|---|---|---|---|---|---|---|"]
for l in lines:
    if not l.startswith("rank"):
        continue
    m = re.match(r"rank (\d+) rows (\d+) nnz own (\d+) other (\d+)", l)
    tab.append("| %s | %s | %s / %s | %s | %s | %s | %s | %s |" % (m.group(1), m.group(2), m.group(3), m.group(4), " / ".join(re.findall(r"fused/\d+us ([\d.]+)", l)),
               " / ".join(re.findall(r"2launch/\d+us ([\d.]+)", l)), re.search(r"step kernel alone ([\d.]+)", l).group(1), re.search(r"own alone ([\d.]+)", l).group(1),
               re.search(r"other alone ([\d.]+)", l).group(1)))
tab.append("")
tab += ["    " + l for l in lines if not l.startswith("rank")]
p = "profiles/r03_multi_gpu_step.md"
s = open(p).read()
a = s.index("| rank | rows | nnz own / other columns |")
b = s.index("\nReading")
if any(l.startswith("max over") for l in lines):          # a probe cut short by its time limit (slow host) leaves the committed table alone
    open(p, "w").write(s[:a] + "\n".join(tab) + "\n" + s[b:])
    print("\n".join(tab[-9:]))
else:
    print("all-ranks probe incomplete (%d ranks): profiles/r03_multi_gpu_step.md keeps its table" % sum(l.startswith("rank") for l in lines))
# all ranks with the direct exchange (loopback) -> profiles/r03_exchange_footprint.md, between the markers
pl = "gpurun_out/round_end_mg_allranks_push.log"
if os.path.exists(pl):
    lines = [l.strip() for l in open(pl) if l.startswith(("rank", "max over", "1-GPU"))]
    if not any(l.startswith("max over") for l in lines):
        lines = []
    tab = ["<!-- allranks-push -->", "## 6. Every rank with the direct exchange (loopback), final batch (`tools/round_end.sh`; fused / two launches, exchange + 0 / 40 us for the links)\n",
           "| rank | fused step 0 / 40 us | two launches 0 / 40 us |", "|---|---|---|"]
    for l in lines:
        if l.startswith("rank"):
            tab.append("| %s | %s | %s |" % (re.match(r"rank (\d+)", l).group(1), " / ".join(re.findall(r"fused/\d+us ([\d.]+)", l)), " / ".join(re.findall(r"2launch/\d+us ([\d.]+)", l))))
    tab.append("")
    tab += ["    " + l for l in lines if not l.startswith("rank")]
    tab.append("<!-- /allranks-push -->")
    p = "profiles/r03_exchange_footprint.md"
    s = open(p).read()
    if not lines:
        pass
    elif "<!-- allranks-push -->" in s:
        s = s[:s.index("<!-- allranks-push -->")] + "\n".join(tab) + s[s.index("<!-- /allranks-push -->") + len("<!-- /allranks-push -->"):]
    else:
        s = s.rstrip("\n") + "\n\n" + "\n".join(tab) + "\n"
    open(p, "w").write(s)

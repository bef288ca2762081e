#!/usr/bin/env python3
"""tools/round_end_collect.py -- after `gpurun -- bash tools/round_end.sh`: turn gpurun_out/round_end_* into the committed profile files
(profiles/traffic.json + r03_traffic.md via traffic_collect.py, r03_hv15r_f64.md, r03_bench_full*.json.log, the all-ranks table of r03_multi_gpu_step.md)."""
import csv, glob, json, os, re, subprocess, sys
ROUND = os.environ.get("ROUND", "r04")
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
os.chdir(root)
if os.path.exists("gpurun_out/round_end_traffic.json"):       # generated on the GPU box, before the bench ran
    import shutil
    shutil.copy("gpurun_out/round_end_traffic.json", "profiles/traffic.json")
    shutil.copy("gpurun_out/round_end_traffic.md", "profiles/%s_traffic.md" % ROUND)
else:
    subprocess.check_call([sys.executable, "tools/traffic_collect.py"], stdout=subprocess.DEVNULL)
rows = list(csv.DictReader(open(glob.glob("gpurun_out/round_end_prof/*/*kernel_stats.csv")[0])))
line = [x for x in open("gpurun_out/round_end_prof.log") if x.startswith('{"metric')][-1].strip()
# r5: the stdout line is the compact driver record; the full record of the same run is the side file (bench.py emit())
d = json.load(open("gpurun_out/round_end_prof_suite.json")) if os.path.exists("gpurun_out/round_end_prof_suite.json") else json.loads(line)
hv = [e for e in json.load(open("profiles/traffic.json")) if e["workload"] == "HV15R"][0]
r = d["roofline"]
out = ["# Round " + ROUND[1:].lstrip("0") + " profile of the bench command -- HV15R stand-in (2 017 169 rows, 275 454 726 nnz, f64), MI355X\n",
       ("Source: `cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -- python3 bench.py --no-suite --no-vendor --steps 20` (kernel trace + stats only; the\n"
       "`--pmc FETCH_SIZE` / `--pmc WRITE_SIZE` passes are separate runs of `dasp_bench HV15R 1 64 20 3`, `profiles/" + ROUND + "_traffic.md`), kernel sources at `kernel_rev %s`\n"
       "(= `profiles/traffic.json`); all of it one batch on one box (`tools/round_end_" + ROUND[:1] + ROUND[2:] + ".sh`, `tools/round_end_collect.py`).  The %s profiled launches include the timed steps behind\n"
       "`roofline.kernel_ms` (r6: HIP events around exactly the run's --steps steps, against the first y allocated), the 200 back-to-back launches of `kernel_ms_separate_launches`, the 200 timed one by one for the spread (`launch_ms_*`), those of the\n"
       "random-values plan and those against the six y candidates (`frac_best_of_n_y`); the profiled average below (%.1f us over all of them) and the bench line's own `kernel_ms` of the same process (%.1f us; single launches\n"
       "min %.1f / p10 %.1f / median %.1f / p90 %.1f / max %.1f, each including the event between two kernels) agree.\n") % (hv["kernel_rev"], rows[0]["Calls"], float(rows[0]["AverageNs"]) / 1e3, r["kernel_ms"] * 1e3, r["launch_ms_min"] * 1e3,
                                            r["launch_ms_p10"] * 1e3, r["launch_ms_median"] * 1e3, r["launch_ms_p90"] * 1e3, r["launch_ms_max"] * 1e3),
       "## rocprofv3 --kernel-trace --stats --output-format csv -- python3 bench.py --no-suite --no-vendor --steps 20\n",
       "| kernel | calls | avg ns | min ns | max ns | % |\n|---|---|---|---|---|---|"]
for q in rows[:5]:
    out.append("| %s | %s | %.0f | %s | %s | %s |" % (q["Name"][:80], q["Calls"], float(q["AverageNs"]), q["MinNs"], q["MaxNs"], q["Percentage"]))
avg = float(rows[0]["AverageNs"])
out.append(("\nRecomputed roofline fraction from the profile: %d B algorithmic / %.0f ns / 8 TB/s = **%.3f** (bench line: %.4f); counter traffic per launch %.4f GB = %.3f x algorithmic\n"
           "(`profiles/" + ROUND + "_traffic.md`: FETCH_SIZE %.4f GB raw x 2 + WRITE_SIZE %.2f MB), i.e. %.2f TB/s of real traffic.\n") %
           (r["algorithmic_bytes_per_launch"], avg, r["algorithmic_bytes_per_launch"] / avg / 8000, r["frac"], hv["traffic_bytes"] / 1e9,
            hv["traffic_bytes"] / r["algorithmic_bytes_per_launch"], hv["fetch_size_bytes_raw"] / 1e9, hv["write_size_bytes"] / 1e6, hv["traffic_bytes"] / avg / 1e3))
out.append("## bench.py's JSON line of the profiled run\n\n```\n%s\n```" % line)
open("profiles/%s_hv15r_f64.md" % ROUND, "w").write("\n".join(out) + "\n")
full = [x for x in open("gpurun_out/round_end_bench.json.log") if x.startswith("{")][-1]
D = json.loads(full)
if os.path.exists("gpurun_out/round_end_bench_suite.json"):      # r5: compact line on stdout (kept as <ROUND>_bench_line.json.log), full record beside it
    open("profiles/%s_bench_line.json.log" % ROUND, "w").write(full)
    D = json.load(open("gpurun_out/round_end_bench_suite.json"))
    full = json.dumps(D) + "\n"
slow = D["roofline"]["frac"] < 0.94                      # the two box populations: ~0.90-0.91 and ~0.97-0.98 on the HV15R headline
open("profiles/%s_bench_full.json.log" % ROUND, "w").write(full)
print("bench record -> %s box: %.4f ms, frac %.4f, traffic/alg %s" % ("slow" if slow else "fast", D["ms_per_step"], D["roofline"]["frac"], D["roofline"].get("traffic_over_algorithmic")))
for s in D["suite"]:
    print("  %-22s %.4f ms frac %.4f rand %.4f pre %.0f devpre %s traffic %s" % (s["workload"], s["event_ms"], s["frac_hbm_roofline"], s.get("frac_hbm_roofline_random_values", 0),
          s["pre_ms"], s.get("pre_ms_device_csr"), s.get("traffic_over_algorithmic")))
# all-ranks tables of the multi-GPU step probes -> profiles/<ROUND>_multi_gpu_step_tables.md (the prose lives in <ROUND>_multi_gpu_step.md)
tabs = ["# " + ROUND + ": every rank of the 8-way partitions, one after the other on ONE MI355X (tools/mg_step_probe.py; direct exchange in loopback, the flags raised N us after the stores)\n"]
for form, tag in (("two-plan fused step (dasp_mg_step_kernel)", "v1"), ("one-stream step (dasp_mg_step2_kernel)", "v2")):
    for w in ("HV15R", "Queen_4147"):
        f = "gpurun_out/round_end_mg_%s_%s.log" % (tag, w)
        if not os.path.exists(f):
            continue
        lines = [l.strip() for l in open(f) if l.startswith(("rank", "max over", "1-GPU"))]
        tabs.append("## %s, %s\n" % (w, form))
        tabs.append("| rank | rows | nnz own / other columns | fused, flags after 0 / 15 / 30 / 45 us | two launches | step kernel alone | plan(s) alone |\n|---|---|---|---|---|---|---|")
        for l in lines:
            if not l.startswith("rank"):
                continue
            m = re.match(r"rank (\d+) rows (\d+) nnz own (\d+) other (\d+)", l)
            alone = re.search(r"own alone ([\d.]+) other alone ([\d.]+)", l)
            alone = ("own %s + other %s" % alone.groups()) if alone else (re.search(r"plain kernel, coarse x / y\) ([\d.]+)", l).group(1) if "plain kernel, coarse" in l else "")
            sk = re.search(r"step kernel alone ([\d.]+)", l)
            tabs.append("| %s | %s | %s / %s | %s | %s | %s | %s |" % (m.group(1), m.group(2), m.group(3), m.group(4), " / ".join(re.findall(r"fused/\d+us ([\d.]+)", l)),
                        " / ".join(re.findall(r"2launch/\d+us ([\d.]+)", l)), sk.group(1) if sk else "", alone))
        tabs.append("")
        tabs += ["    " + l for l in lines if not l.startswith("rank")]
        tabs.append("")
open("profiles/%s_multi_gpu_step_tables.md" % ROUND, "w").write("\n".join(tabs) + "\n")
print("multi-GPU tables -> profiles/%s_multi_gpu_step_tables.md" % ROUND)

#!/usr/bin/env python3
"""tools/traffic.py <tag> <workload> <precision> <scale> -> one profiles/traffic.json entry (stdout, JSON) + a markdown summary (stderr)
from the three rocprofv3 passes tools/prof.sh wrote under gpurun_out/prof_<tag>/ for `dasp_bench <workload> <scale> <precision> ...`:
kernel trace + stats, --pmc FETCH_SIZE, --pmc WRITE_SIZE (separate passes: the two do not fit one pass on gfx950).
HBM bytes per SpMV = sum over every dasp_* kernel of one SpMV (column panels: P fused kernels + the panel sum; long rows: + stage 2) of
2 x FETCH_SIZE (gfx950 tallies the 128-byte requests of wide streaming reads at 64 B: MI355X_MICROARCH.md, HBM) + WRITE_SIZE.
The x2 is calibrated for wide coalesced streams only: for gather-bound kernels the entry says so and also carries the raw sum."""
import os
import csv, glob, hashlib, json, os, sys
tag, workload, prec, scale = sys.argv[1], sys.argv[2], int(sys.argv[3]), float(sys.argv[4])
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
d = os.path.join(root, "gpurun_out", "prof_" + tag)


def rev():
    h = hashlib.sha1()
    for f in ("kernels.hip", "spmv_device.hpp", "upload.cpp", "plan.cpp", "device.hpp", "plan.hpp", "twophase.cpp"):      # = bench.py kernel_revision()
        h.update(open(os.path.join(root, "dasp_amd", "csrc", f), "rb").read())
    return h.hexdigest()[:12]


SPMV_KERNELS = ("dasp_spmv_panels_kernel", "dasp_long_reduce_panels_kernel", "dasp_lcb_reduce_kernel", "dasp_lcb_kernel", "dasp_spmv_kernel", "dasp_long_reduce_kernel", "dasp_panel_sum_kernel",
                "dasp_tp_expand_kernel", "dasp_tp_reduce_kernel")   # the kernels of one SpMV (not dasp_bench's packers); longer names first: matched by substring


def spmv_kernel(name):
    """which SpMV kernel a (possibly mangled: rocprofv3 does not demangle the _Float16 instantiations) name is, or None"""
    if "dasp_spmv_win1_kernel" in name or "dasp_spmv_rt_kernel" in name:      # the 128-register build of the windowed kernel (plans of <= 256 windows) / a column panel with row tiles
        return "dasp_spmv_kernel"
    for k in SPMV_KERNELS:
        if k in name:
            return k
    return None


def per_kernel(kind):
    rows = list(csv.DictReader(open(max(glob.glob(d + "/pmc_%s/*/*counter_collection.csv" % kind), key=os.path.getmtime))))      # the newest pass
    tot, cnt = {}, {}
    for r in rows:
        k = spmv_kernel(r["Kernel_Name"])
        if k is None:
            continue
        tot[k] = tot.get(k, 0.0) + float(r["Counter_Value"]) * 1024
        cnt[k] = cnt.get(k, 0) + 1
    return tot, cnt, rows


fetch, fcnt, frows = per_kernel("fetch")
write, wcnt, _ = per_kernel("write")
log = open(d + "/trace.log").read()
panels = 0
for tok in log.replace("|", " ").split():
    if tok.startswith("panels="):
        panels = int(tok.split("=")[1])
two_phase = "dasp_spmv_kernel" not in fcnt and "dasp_tp_reduce_kernel" in fcnt          # a two-phase plan: one expand + one reduce launch per SpMV
merged = "dasp_spmv_panels_kernel" in fcnt                                              # r5: the panels of a column-panel plan in one launch
main_k = "dasp_tp_reduce_kernel" if two_phase else "dasp_spmv_panels_kernel" if merged else "dasp_spmv_kernel"
per_spmv = 1 if (two_phase or merged) else max(1, panels)
n_spmv = fcnt[main_k] / per_spmv
f_raw = sum(fetch.values()) / n_spmv
w = sum(write.values()) / (wcnt[main_k] / per_spmv)
stats = list(csv.DictReader(open(max(glob.glob(d + "/trace/*/*kernel_stats.csv"), key=os.path.getmtime))))
sys.stderr.write("## rocprofv3 --kernel-trace --stats -- dasp_bench %s %g %d (tag %s)\n\n| kernel | calls | avg ns | %% |\n|---|---|---|---|\n" % (workload, scale, prec, tag))
for r in stats[:5]:
    sys.stderr.write("| %s | %s | %.0f | %s |\n" % (r["Name"][:90], r["Calls"], float(r["AverageNs"]), r["Percentage"]))
g = [r for r in frows if ("dasp_tp_reduce" if two_phase else "dasp_spmv") in r["Kernel_Name"]][0]
sys.stderr.write("\nVGPR_Count=%s SGPR_Count=%s LDS=%s scratch=%s workgroup=%s; %d column panels; %.0f SpMVs profiled\n" %
                 (g["VGPR_Count"], g["SGPR_Count"], g["LDS_Block_Size"], g["Scratch_Size"], g["Workgroup_Size"], panels, n_spmv))
sys.stderr.write("FETCH_SIZE (own pass) = %.4f GB raw per SpMV -> x2 = %.4f GB; WRITE_SIZE (own pass) = %.2f MB; traffic = %.4f GB per SpMV\n\n" %
                 (f_raw / 1e9, 2 * f_raw / 1e9, w / 1e6, (2 * f_raw + w) / 1e9))
avg = {}
for r in stats:
    k = spmv_kernel(r["Name"])
    if k:
        avg[k] = avg.get(k, 0.0) + float(r["AverageNs"]) * (max(1, panels) if k == "dasp_spmv_kernel" and not merged else 1)   # per SpMV
print(json.dumps({"workload": workload, "precision": prec, "scale": scale, "kernel_rev": rev(), "kernels": sorted(fcnt),
                  "fetch_size_bytes_raw": round(f_raw), "write_size_bytes": round(w), "traffic_bytes": round(2 * f_raw + w),
                  "kernel_avg_ns": avg, "column_panels": panels, "two_phase": bool(two_phase),
                  "correction": "2 x FETCH_SIZE (gfx950, MI355X_MICROARCH.md HBM section; calibrated for wide coalesced streams) + WRITE_SIZE, summed over the kernels of one SpMV",
                  "source": "profiles/%s_traffic.md" % os.environ.get("ROUND", "r04")}))

#!/usr/bin/env python3
"""tools/summarize_prof.py <gpurun_out/prof_TAG> <command string> -> markdown summary on stdout (+ traffic numbers).
Reads the three rocprofv3 passes tools/prof.sh wrote (kernel trace + stats, --pmc FETCH_SIZE, --pmc WRITE_SIZE)."""
import csv, glob, sys
d, cmd = sys.argv[1], sys.argv[2]
print(f"## rocprofv3 --kernel-trace --stats --output-format csv -- {cmd}\n")
def pick(pattern):
    """the CSV of the process that ran the DASP kernels (bench.py's vendor comparator is a child process with its own files)"""
    files = sorted(glob.glob(pattern))
    for f in files:
        if "dasp_spmv" in open(f).read():
            return f
    return files[0]
rows = list(csv.DictReader(open(pick(d + "/trace/*/*kernel_stats.csv"))))
print("| kernel | calls | avg ns | min ns | max ns | % |\n|---|---|---|---|---|---|")
for r in rows[:4]:
    print(f"| {r['Name'][:80]} | {r['Calls']} | {float(r['AverageNs']):.0f} | {r['MinNs']} | {r['MaxNs']} | {r['Percentage']} |")
res = {}
for kind, cname in (("fetch", "FETCH_SIZE"), ("write", "WRITE_SIZE")):
    rows = list(csv.DictReader(open(pick(d + f"/pmc_{kind}/*/*counter_collection.csv"))))
    v = [float(r["Counter_Value"]) for r in rows if "dasp_spmv" in r["Kernel_Name"]]
    res[cname] = sum(v) / len(v) * 1024
    g = [r for r in rows if "dasp_spmv" in r["Kernel_Name"]][0]
    print(f"\n`--pmc {cname}` (own pass): mean over {len(v)} dispatches of dasp_spmv_kernel = {sum(v)/len(v):.1f} KB (min {min(v):.1f}, max {max(v):.1f})")
print(f"\nVGPR_Count={g['VGPR_Count']} Accum_VGPR_Count={g['Accum_VGPR_Count']} SGPR_Count={g['SGPR_Count']} LDS={g['LDS_Block_Size']} scratch={g['Scratch_Size']} workgroup={g['Workgroup_Size']} grid={g['Grid_Size']}")
print(f"\nFETCH_SIZE = {res['FETCH_SIZE']/1e9:.4f} GB raw -> x2 (gfx950: the counter tallies 128-B requests at 64 B) = {2*res['FETCH_SIZE']/1e9:.4f} GB; WRITE_SIZE = {res['WRITE_SIZE']/1e6:.2f} MB; traffic = {(2*res['FETCH_SIZE']+res['WRITE_SIZE'])/1e9:.4f} GB per launch")
print(f"TRAFFIC_BYTES={2*res['FETCH_SIZE']+res['WRITE_SIZE']:.0f} FETCH_RAW={res['FETCH_SIZE']:.0f} WRITE={res['WRITE_SIZE']:.0f}")

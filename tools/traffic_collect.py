#!/usr/bin/env python3
"""tools/traffic_collect.py -- after tools/traffic_all.sh ran on the GPU box: turn every gpurun_out/prof_<tag>/ into profiles/traffic.json
(one entry per workload, stamped with the kernel revision) and profiles/<ROUND>_traffic.md (summary table + the per-workload rocprofv3 summaries)."""
import json, os, subprocess, sys
ROUND = os.environ.get("ROUND", "r04")
import numpy as np
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, root)
LIST = [("hv15r64", "HV15R", 64), ("cop64", "cop20k_A", 64), ("nlp64", "nlpkkt160", 64), ("pl64", "powerlaw_1M", 64), ("queen64", "Queen_4147", 64),
        ("wb16", "webbase-1M", 16), ("lj16", "ljournal-2008", 16), ("rmat16", "rmat_2M", 16), ("lju16", "ljournal-2008-uniform", 16),
        ("hvu64", "HV15R-unstructured", 64), ("wbu16", "webbase-1M-uniform", 16)]
import dasp_amd as D
entries, mds = [], []
for tag, w, prec in LIST:
    r = subprocess.run([sys.executable, os.path.join(root, "tools", "traffic.py"), tag, w, str(prec), "1"], capture_output=True, text=True)
    if r.returncode != 0:
        sys.exit("traffic.py failed for %s:\n%s" % (tag, r.stderr[-2000:]))
    entries.append(json.loads(r.stdout)); mds.append(r.stderr)
json.dump(entries, open(os.path.join(root, "profiles", "traffic.json"), "w"), indent=1)
out = ["# " + ROUND + ": HBM traffic per SpMV from rocprofv3 PMC passes (kernels of this round, kernel_rev %s)\n" % entries[0]["kernel_rev"],
       "Collected with `tools/traffic_all.sh` on the GPU box (per workload: `tools/prof.sh <tag> -- dasp_amd/bin/dasp_bench <workload> 1 <precision> <iters> 3`: "
       "kernel trace + stats, `--pmc FETCH_SIZE`, `--pmc WRITE_SIZE` in separate passes), turned into `profiles/traffic.json` and this file by "
       "`tools/traffic_collect.py`.  traffic = 2 x FETCH_SIZE + WRITE_SIZE summed over the kernels of one SpMV (the x2: gfx950 tallies the 128-byte requests of wide "
       "streaming reads at 64 B, MI355X_MICROARCH.md HBM section; calibrated for wide coalesced streams, so the gather-bound rows over-state their reads).  "
       "B_alg = nnz (vb + 4) + (m + 1) 4 + (n + m) vb.\n",
       "| workload | dtype | B_alg GB | FETCH_SIZE raw GB | WRITE_SIZE MB | traffic GB | traffic / B_alg | kernels of one SpMV, us (rocprof avg) | traffic / time, TB/s |",
       "|---|---|---|---|---|---|---|---|---|"]
for (tag, w, prec), e in zip(LIST, entries):
    rp, ci = None, None
    m, n = D.synth_dims(w, 1.0)
    nnz = int(np.asarray(D.synth_row_lengths(w, 1.0), dtype=np.int64).sum())
    vb = 8 if prec == 64 else 2
    balg = nnz * (vb + 4) + (m + 1) * 4 + (n + m) * vb
    us = sum(e["kernel_avg_ns"].values()) / 1e3
    out.append("| %s | f%d | %.4f | %.4f | %.2f | %.4f | %.3f | %.1f | %.2f |" % (w, prec, balg / 1e9, e["fetch_size_bytes_raw"] / 1e9, e["write_size_bytes"] / 1e6,
               e["traffic_bytes"] / 1e9, e["traffic_bytes"] / balg, us, e["traffic_bytes"] / us / 1e6))
open(os.path.join(root, "profiles", ROUND + "_traffic.md"), "w").write("\n".join(out) + "\n\n" + "\n".join(mds))
print("\n".join(out[2:]))

#!/usr/bin/env python3
"""tools/r6_mfma_table.py -- gpurun_out/pmc_mfma_*.txt / pmc_tp_lj16.txt (written by tools/round_end_r6.sh through tools/pmc.sh) -> profiles/r06_mfma_util.md"""
import os, re
root = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out") + "/"
rows = [("HV15R f64 (`dasp_spmv_kernel<double,1,1,0,1,0>`)", "pmc_mfma_hv15r64.txt", "F64"), ("nlpkkt160 f64 (`dasp_spmv_kernel<double,1,1,0,0,7>`: the 7-wave build)", "pmc_mfma_nlp64.txt", "F64"),
        ("ljournal-2008 f16, DASP form (`two_phase = -1`: 4 column panels in one launch, `dasp_spmv_panels_kernel<half,..>`)", "pmc_mfma_lj16_dasp.txt", "F16"),
        ("rmat_2M f16, DASP form (`two_phase = -1`, `dasp_spmv_kernel<half,..>`)", "pmc_mfma_rmat16.txt", "F16"), ("webbase-1M f16", "pmc_mfma_wb16.txt", "F16"),
        ("ljournal-2008 f16, two-phase form (the product's choice; `dasp_tp_expand` + `dasp_tp_reduce`, mean of the two dispatches)", "pmc_tp_lj16.txt", "F16")]
out = ["# r06 -- MFMA utilisation of this round's kernels at full size (north_star: \"choices evidenced by rocprof HBM GB/s and MFMA utilisation\"; as in r5)\n",
       "Collected by `tools/round_end_r6.sh` (`tools/pmc.sh`: `rocprofv3 --kernel-trace --pmc <one group per pass> -- dasp_amd/bin/dasp_bench <workload> 1 <precision> ...`, the program itself after `--`); raw summaries `gpurun_out/pmc_mfma_*.txt`; this table: `tools/r6_mfma_table.py`.",
       "Counters are sums over the device (8 XCDs, 256 CUs, 1024 SIMDs); per dispatch means.  MFMA pipe busy = `SQ_VALU_MFMA_BUSY_CYCLES` / 1024 SIMDs / (`GRBM_GUI_ACTIVE` / 8 XCDs).  HBM side: `profiles/r05_traffic.md`.\n",
       "| workload (kernel) | dispatch us (under PMC) | MFMA instructions | MFMA MOPS (F64 / F16) | MFMA busy cycles | GPU-active cycles per XCD | **MFMA pipe busy** | VALU instructions | waves |", "|---|---|---|---|---|---|---|---|---|"]
for name, f, kind in rows:
    t = open(root + f).read()
    g = lambda k: float(re.search(k + r"\s+mean ([0-9.e+]+)", t).group(1))
    us = float(re.search(r"median ([0-9.]+) us", t).group(1))
    busy, act = g("SQ_VALU_MFMA_BUSY_CYCLES"), g("GRBM_GUI_ACTIVE") / 8
    out.append("| %s | %.1f | %.3g | %.3g | %.3g | %.3g | **%.1f %%** | %.3g | %d |" % (name, us, g("SQ_INSTS_MFMA"), g("SQ_INSTS_VALU_MFMA_MOPS_" + kind), busy, act, 100 * busy / 1024 / act, g("SQ_INSTS_VALU"), g("SQ_WAVES")))
out += ["", "Reading:",
        "* The HBM-bound f64 kernels keep the matrix pipe **23-28 % busy** while streaming at 0.87-0.98 of the 8 TB/s roofline: one `v_mfma_f64_16x16x4_f64` (64 SIMD cycles) per 64 stored elements;",
        "  with the DASP diagonal trick 1/16 of its 2048 flops is the row's dot product, so HV15R's 4.34 M MFMAs per SpMV are 8.9 GFLOP of matrix-pipe work (20.6 TFLOP/s) for 0.55 GFLOP of SpMV.  MFMA is the reduction engine, not the bound (DESIGN.md section 4) -- at 6 waves per SIMD the pipe has 3.6x headroom.",
        "* On the gather-bound f16 graphs the pipe is idle (**0.3-0.8 %**): the waves wait for x.  That is the evidence behind this round's two-phase form for such matrices: it uses no MFMA at all",
        "  (f16 x f16 products in the VALU, LDS f64 atomics) and halves the time (`profiles/r05_two_phase.md`); the column-blocked long rows of powerlaw_1M (`profiles/r05_long_cb.md`) are plain FMAs on LDS-staged x for the same reason.",
        "* Same kernels as r5 for these rows (the r6 changes are in the window kernels, the long pieces' ids and the f16 grid order): the figures match `profiles/r05_mfma_util.md`."]
open(os.path.join(os.path.dirname(root.rstrip("/")), "profiles", "r06_mfma_util.md"), "w").write("\n".join(out) + "\n")
print("\n".join(out[3:11]))

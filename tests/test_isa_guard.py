"""Performance cliffs that DESIGN.md states in terms of the compiled code, turned into a guard: the invariants
below were found by reading listings by hand; an innocent edit (a fence, an atomic, a wave barrier in the wrong place) or a compiler bump flips
them without any test noticing -- HV15R-unstructured in two forced panels 0.555 -> 0.884 ms when __builtin_amdgcn_wave_barrier() sits in
row_tile (302 -> 119 s_load), the own-column product of an 8-way HV15R slice 61 -> 92 us with an atomic store in put_y.

Compile-only: reads the gfx950 code objects inside dasp_amd/csrc/build/*.o (tools/isa_report.py: metadata note + disassembly), no GPU."""
import importlib.util
import os

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def isa():
    import __graft_entry__ as g
    g.build()                                                   # the objects of THIS tree (no-op when they are up to date)
    spec = importlib.util.spec_from_file_location("isa_report", os.path.join(ROOT, "tools", "isa_report.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    rows = m.report()
    assert len(rows) >= 40, sorted(rows)
    return rows


PLAIN64 = ["dasp_spmv_kernel<double,%d,%d,0,%d,0,0>" % (nt, c16, c8) for nt in (0, 1) for c16, c8 in ((0, 0), (1, 0), (1, 1))]
SEVEN64 = [k[:-5] + ",7,0>" for k in PLAIN64]      # r5: the same held to 7 waves per SIMD, for plans of one-shot blocks (DevicePlan::seven_waves)
PLAIN16 = ["dasp_spmv_kernel<half,%d,%d,0,0,0,0>" % (nt, c16) for nt in (0, 1) for c16 in (0, 1)]
# r6: the builds that read the 16-bit ids of narrow long pieces (DevicePlan::long16): the plain kernels' budget
LONG16_64 = ["dasp_spmv_kernel<double,%d,%d,0,0,0,1>" % (nt, c16) for nt in (0, 1) for c16 in (0, 1)]
LONG16_16 = ["dasp_spmv_kernel<half,%d,%d,0,0,0,1>" % (nt, c16) for nt in (0, 1) for c16 in (0, 1)]
RT64 = ["dasp_spmv_rt_kernel<double,%d,%d>" % (nt, c16) for nt in (0, 1) for c16 in (0, 1)]
RT16 = ["dasp_spmv_rt_kernel<half,%d,%d>" % (nt, c16) for nt in (0, 1) for c16 in (0, 1)]
WIN = ["dasp_spmv_kernel<%s,%d,%d,1,0,0,0>" % (t, nt, c16) for t in ("double", "half") for nt in (0, 1) for c16 in (0, 1)]
WIN1 = ["dasp_spmv_win1_kernel<%s,%d>" % (t, c16) for t in ("double", "half") for c16 in (0, 1)]
STEP = ["dasp_mg_step_kernel<0>", "dasp_mg_step_kernel<1>", "dasp_mg_step2_kernel<0>", "dasp_mg_step2_kernel<1>"]


def test_every_kernel_of_the_hot_path_is_there(isa):
    for k in PLAIN64 + SEVEN64 + PLAIN16 + LONG16_64 + LONG16_16 + RT64 + RT16 + WIN + WIN1 + STEP + ["dasp_long_reduce_kernel<double>", "dasp_long_reduce_kernel<half>",
                                                                    "dasp_panel_sum_kernel<double,2>", "dasp_panel_sum_kernel<half,8>"]:
        assert k in isa, k
    # the MFMA kernels issue MFMAs, and agree on wave64 geometry: 4 waves of 64 (256) or a window workgroup (1024)
    for k in PLAIN64 + PLAIN16 + RT64 + RT16 + WIN + WIN1 + STEP:
        assert isa[k]["mfma"] > 0 and isa[k].get("agpr_count", 0) == 0, (k, isa[k])      # -amdgpu-mfma-vgpr-form: accumulators stay in VGPRs


def test_plain_kernels_have_no_scratch_and_keep_their_occupancy(isa):
    """DESIGN.md 4.4 / kernels.hip: the non-windowed kernels never spill; f64 <= 80 VGPRs (6 waves per SIMD), f16 <= 72 (7 waves; the row-tile
    build is held there by its launch bound)"""
    for k in PLAIN64 + RT64 + LONG16_64:
        r = isa[k]
        # no scratch INSTRUCTION and no spilled VGPR.  (r5: the 32-bit-id row-tile build reports a 36-byte private segment that nothing accesses -- the frame slots of nine
        # SGPRs spilled to VGPR lanes at the 104-SGPR ceiling; the product reaches that code through dasp_spmv_panels_kernel, which has none.)
        assert r["vgpr_spill_count"] == 0 and r["scratch"] == 0 and r["private_segment_fixed_size"] <= (64 if k in RT64 else 0), (k, r)
        assert r["vgpr_count"] <= 80, (k, r["vgpr_count"])
    for k in PLAIN16 + RT16 + LONG16_16:
        r = isa[k]
        assert r["private_segment_fixed_size"] == 0 and r["vgpr_spill_count"] == 0 and r["scratch"] == 0, (k, r)
        assert r["vgpr_count"] <= 72, (k, r["vgpr_count"])
    for k in SEVEN64:       # 72 registers; what that costs the pipelined path (not what these builds are launched for) is a few spilled dwords: growth is a regression
        r = isa[k]
        # (r6: the one-byte-id build of the two went from 12 to 20 bytes / 8 to 17 scratch instructions in its pipelined path when the blocks' tail-step count became a scalar table load)
        assert r["vgpr_count"] <= 72 and r["private_segment_fixed_size"] <= 24 and r["scratch"] <= 20, (k, r["vgpr_count"], r["private_segment_fixed_size"], r["scratch"])


def test_row_tables_stay_scalar_loads(isa):
    """spmv_device.hpp tab<> / row_tile / put_y: the per-block tables (med_ptr, med_base per chunk, med_c8ptr, piece_ptr ...) are fetched with
    s_load.  With a fence, a wave barrier or an atomic anywhere in the function the compiler turns every one of them into a vector load:
    the 16-bit-id f64 kernels then drop from ~300 to ~120 s_load instructions (and HV15R-class matrices lose a third of their speed)."""
    for k in PLAIN64 + RT64:
        c16 = k.split(",")[2][0] == "1"
        # (r5: the plan's arguments come from a device-resident block loaded once, not from kernarg fields re-read at every use -- ~85 fewer s_load
        # than the by-value builds of r4: 302 -> 217 / 116 -> 31.  What is left in a 16-bit-id build are the per-chunk table loads the cliff is about.)
        assert isa[k]["s_load"] >= (200 if c16 else 25), (k, isa[k]["s_load"])
    # the row-tile build must not lose scalar loads against the kernel without tiles (the f64 panel kernel is the one the cliff was found on)
    for nt in (0, 1):
        for c16 in (0, 1):
            assert isa["dasp_spmv_rt_kernel<double,%d,%d>" % (nt, c16)]["s_load"] >= isa["dasp_spmv_kernel<double,%d,%d,0,0,0,0>" % (nt, c16)]["s_load"] - 40      # (how loads merge differs by a few dozen)
            assert isa["dasp_spmv_rt_kernel<half,%d,%d>" % (nt, c16)]["s_load"] >= isa["dasp_spmv_kernel<half,%d,%d,0,0,0,0>" % (nt, c16)]["s_load"] - 8
    for k in PLAIN16 + RT16:
        assert isa[k]["s_load"] >= 25, (k, isa[k]["s_load"])
    # pointers of the device-resident argument block must be known to be GLOBAL (ldp in load_args): as flat pointers every table and tile load becomes
    # flat_load and none of them scalar (f64 16-bit-id kernel: 1056 global_load + 20 flat stores -> 297 + 972)
    # (the windowed kernels hold the short-tile code twice since r6 -- as fillers of the window workgroups and for the plans whose tiles keep workgroups of their own --
    # and with it twice its x gathers / y stores through the generic x / y pointers: 90 flat ops, none of them a table or tile load)
    for k in PLAIN64 + RT64 + PLAIN16 + RT16 + WIN + WIN1:
        assert isa[k]["flat"] <= (100 if k in WIN + WIN1 else 80) and isa[k]["global_load"] >= 400, (k, isa[k]["flat"], isa[k]["global_load"])
        assert isa[k]["kernarg_segment_size"] <= 64, (k, isa[k]["kernarg_segment_size"])
    # the multi-GPU step kernels wait on flags in memory; their tables go through the constant address space (tab<true>) and must stay scalar
    assert isa["dasp_mg_step_kernel<1>"]["s_load"] >= 200 and isa["dasp_mg_step2_kernel<1>"]["s_load"] >= 700, [isa[k]["s_load"] for k in STEP]


def test_windowed_kernels_keep_their_register_cap_and_known_scratch(isa):
    """the windowed kernel is held to 64 VGPRs so that two 1024-thread window workgroups share a CU (12.9 vs 15.0 us on cop20k_A); what that
    costs in scratch is an accepted, recorded figure -- growth is a regression.  The one-window-per-CU build (win1) has no cap and no scratch."""
    # r6: the f64 builds run their blocks in batches of 3 (one shot up to 6 steps) instead of the plain kernels' 4 / 8 and no longer spill
    # (60 / 40 bytes of scratch, 12-29 VGPRs in r5; cop20k_A x16 114.1 -> 103.1 us, x64 438 -> 413); the f16 build with 32-bit ids keeps its two spilled registers
    accepted = {"dasp_spmv_kernel<double,0,0,1,0,0,0>": 0, "dasp_spmv_kernel<double,0,1,1,0,0,0>": 0, "dasp_spmv_kernel<half,0,0,1,0,0,0>": 12, "dasp_spmv_kernel<half,0,1,1,0,0,0>": 0}
    for k in WIN:
        r = isa[k]
        assert r["vgpr_count"] <= 64, (k, r["vgpr_count"])
        assert r["private_segment_fixed_size"] <= accepted[k.replace(",1,", ",0,", 1) if k.split(",")[1] == "1" else k], (k, r["private_segment_fixed_size"])
        assert r["s_barrier"] == 1 and r["ds"] > 0                                          # ONE barrier: behind the x copy
        if k.startswith("dasp_spmv_kernel<double"):
            assert r["vgpr_spill_count"] == 0 and r["scratch"] == 0, (k, r)
    for k in WIN1:
        r = isa[k]
        assert r["private_segment_fixed_size"] == 0 and r["vgpr_count"] <= 128 and r["s_barrier"] == 1, (k, r)


def test_step_kernels_fit_their_occupancy(isa):
    """mgstep.hip: <= 80 VGPRs (6 waves per SIMD; 83-85 with the wave-segmented short rows compiled in, which is why they are not), no scratch"""
    for k in STEP:
        r = isa[k]
        assert r["vgpr_count"] <= 80 and r["private_segment_fixed_size"] == 0 and r["scratch"] == 0, (k, r)


def test_row_tile_lds_ops_are_in_program_order(isa):
    """ADVICE r4 (spmv_device.hpp row_tile): lane i parks products in LDS, other lanes sum them, with NO fence / wave barrier between (either costs
    the scalar loads above).  Correct because one wave's LDS operations execute in issue order -- as long as the compiler emits every ds_write of
    the store loop BEFORE the first ds_read of the sum loop.  Checked on the listing: in the tile section of every row-tile kernel the writes
    precede the reads."""
    import re
    import subprocess
    import tempfile
    spec = importlib.util.spec_from_file_location("isa_report", os.path.join(ROOT, "tools", "isa_report.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    with tempfile.TemporaryDirectory() as work:
        co = m.code_object(os.path.join(ROOT, "dasp_amd", "csrc", "build", "kernels.o"), work)
        txt = subprocess.run([os.path.join(m.LLVM, "llvm-objdump"), "-d", co], capture_output=True, text=True, check=True).stdout
    seen = 0
    for sec in re.split(r"\n(?=[0-9a-f]+ <)", txt):
        head = sec.split("\n", 1)[0]
        if "dasp_spmv_rt_kernel" not in head:
            continue
        ops = [(i, l) for i, l in enumerate(sec.splitlines()) if re.search(r"\tds_(write|read|store|load)", l)]
        writes = [i for i, l in ops if re.search(r"ds_(write|store)", l)]
        reads = [i for i, l in ops if re.search(r"ds_(read|load)", l)]
        assert writes and reads, head
        # the store loop is unrolled by 4 (+ remainder): every write of the LAST store-loop body lies before the first read
        assert max(writes) < min(reads), (head, writes, reads)
        seen += 1
    assert seen == 8


def test_round_5_kernels(isa):
    """the kernels added in r5: the merged-panel launch keeps the plain kernels' budget (<= 80 VGPRs, no scratch, scalar table loads); the two-phase and
    column-blocked kernels stream with global (not flat) loads, use no scratch and no MFMA, and synchronise exactly where the design says (x slice staged /
    accumulators ready)"""
    for nt in (0, 1):
        for c16 in (0, 1):
            for t, cap in (("double", 80), ("half", 72)):
                r = isa["dasp_spmv_panels_kernel<%s,%d,%d>" % (t, nt, c16)]
                assert r["vgpr_count"] <= cap and r["private_segment_fixed_size"] == 0 and r["scratch"] == 0 and r["flat"] == 0, r
            assert isa["dasp_spmv_panels_kernel<double,%d,%d>" % (nt, c16)]["s_load"] >= (200 if c16 else 25)
    for k in ("dasp_tp_expand_kernel<half>", "dasp_tp_reduce_kernel<half>", "dasp_lcb_kernel<double>", "dasp_lcb_kernel<half>"):
        r = isa[k]
        assert r["private_segment_fixed_size"] == 0 and r["scratch"] == 0 and r["mfma"] == 0 and r["flat"] == 0 and r["ds"] > 0, (k, r)
    assert isa["dasp_tp_expand_kernel<half>"]["s_barrier"] == 1 and isa["dasp_tp_reduce_kernel<half>"]["s_barrier"] == 2
    assert isa["dasp_lcb_kernel<double>"]["s_barrier"] == 2 and isa["dasp_lcb_kernel<double>"]["vgpr_count"] <= 128      # 1024 threads: 4 waves per SIMD
    assert isa["dasp_lcb_kernel<half>"]["vgpr_count"] <= 128

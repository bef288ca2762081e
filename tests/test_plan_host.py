"""Host half of the plan (classifier + packers, no GPU): category counts and the output
permutation are bit-identical to the oracle's restatement of the reference; the native packed
arrays decode back to exactly the CSR rows (every nonzero once, in row order, pads only at row ends)."""
import os

import numpy as np
import pytest

import util

CASES = [
    ("mixed", util.mixed_matrix, 3000, 2500, 7),
    ("pairs", util.pair_heavy_matrix, 4000, 3000, 11),
    ("tiny", util.mixed_matrix, 37, 50, 3),
]
COUNT_FIELDS = "row_long row_block row_zero rowloop short_row_1 short_row_2 short_row_3 short_row_4 common_13 nnz_short nnz_long".split()   # nnz_irreg / origin_nnz_reg depend on the tile geometry


@pytest.mark.parametrize("prec", [64, 16])
@pytest.mark.parametrize("tag,builder,m,n,seed", CASES)
def test_counts_and_order_match_oracle(dasp, oracle, prec, tag, builder, m, n, seed):
    dt = np.float64 if prec == 64 else np.float16
    rp, ci, v = builder(m, n, seed, values="f16" if prec == 16 else "uniform", dtype=dt)
    plan = dasp.Plan(rp, ci, v, n, precision=prec)
    P = oracle.Packed(prec, rp, ci, v.astype(np.float64), n)
    st = plan.stats
    for f in COUNT_FIELDS:
        assert st[f] == getattr(P, f), f
    assert (plan.order_rid == P.order_rid).all()          # identical output permutation
    assert st["data_origin1"] == (rp[-1] + n + m) * (prec // 8) + rp[-1] * 4 + (m + 1) * 4   # main_f64.cu:143


@pytest.mark.parametrize("prec", [64, 16])
@pytest.mark.parametrize("tag,builder,m,n,seed", CASES)
@pytest.mark.parametrize("piece,seg", [(0, 0), (256, 0), (0, 1), (0, -1)])
def test_native_format_decodes_to_csr(dasp, prec, tag, builder, m, n, seed, piece, seg):
    dt = np.float64 if prec == 64 else np.float16
    rp, ci, v = builder(m, n, seed, values="f16" if prec == 16 else "uniform", dtype=dt)
    plan = dasp.Plan(rp, ci, v, n, precision=prec, long_piece=piece, short_seg=seg)      # short rows: slabs / wave-segmented (auto: f64 segmented, f16 slabs)
    rows = util.decode_plan(plan)
    order = plan.order_rid
    assert sorted(rows) == list(range(m))
    for slot in range(m):
        r = order[slot]
        cs, vs = rows[slot]
        assert cs == ci[rp[r]:rp[r + 1]].tolist(), (slot, r)
        assert np.array_equal(np.asarray(vs, dt), v[rp[r]:rp[r + 1]]), (slot, r)
    st = plan.stats
    stored = st["fill0_nnz_short"] + st["fill0_nnz_long"] + st["fill0_nnz_reg"] + st["nnz_irreg"]
    assert stored >= st["nnzA"] and abs(st["rate_fill0"] - (stored - st["nnzA"]) / max(st["nnzA"], 1)) < 1e-12


REF_FIELDS = "fill0_nnz_short fill0_nnz_long fill0_nnz_reg nnz_irreg origin_nnz_reg blocknum warp_number data_X".split()


@pytest.mark.parametrize("prec", [64, 16])
@pytest.mark.parametrize("tag,builder,m,n,seed", CASES + [("pairs_big", util.pair_heavy_matrix, 70000, 5000, 13)])
@pytest.mark.parametrize("opts", [{}, {"threshold": 0.5, "block_longest": 64}, {"x_window": 81920, "row_window": 128}, {"col_panels": 2}, {"slab_max_len": 12, "piece_min_len": 100}])
def test_reference_geometry_stats_match_the_oracle(dasp, oracle, prec, tag, builder, m, n, seed, opts):
    """dasp_stats_t::ref_* -- the padded sizes, rate_fill0 and data_X the CUDA reference computes for this input (8-row blocks, 8 x 4
    tiles; dasp_f64.h:609-629,1000-1014,1044-1091,1159-1166 and their f16 counterparts) -- equal the oracle's reference-geometry packer
    whatever native layout the plan itself chose (windows, column panels, slabs, pieces): they are what dasp_spmv_all_* writes into
    the reference's CSV columns."""
    dt = np.float64 if prec == 64 else np.float16
    rp, ci, v = builder(m, n, seed, values="f16" if prec == 16 else "uniform", dtype=dt)
    plan = dasp.Plan(rp, ci, v, n, precision=prec, **opts)
    P = oracle.Packed(prec, rp, ci, v.astype(np.float64), n, threshold=opts.get("threshold", 0.75), block_longest=opts.get("block_longest", 256))
    st = plan.stats
    for f in REF_FIELDS:
        assert st["ref_" + f] == getattr(P, f), f
    assert abs(st["ref_rate_fill0"] - P.rate_fill0) < 1e-12
    plan.close()


def test_threshold_and_block_longest_change_the_split(dasp):
    rp, ci, v = util.mixed_matrix(2000, 1500, 9)
    a = dasp.Plan(rp, ci, v, 1500, threshold=0.75).stats
    b = dasp.Plan(rp, ci, v, 1500, threshold=0.25).stats
    assert b["nnz_irreg"] <= a["nnz_irreg"] and b["fill0_nnz_reg"] >= a["fill0_nnz_reg"]
    c = dasp.Plan(rp, ci, v, 1500, block_longest=64).stats
    assert c["row_long"] > a["row_long"] and c["row_block"] < a["row_block"]


def test_empty_and_degenerate(dasp):
    for m, n in [(0, 0), (5, 3)]:
        rp = np.zeros(m + 1, np.int32)
        plan = dasp.Plan(rp, np.zeros(0, np.int32), np.zeros(0), n)
        assert plan.stats["row_zero"] == m and plan.order_rid.tolist() == list(range(m))


def test_column_remap_for_partitioned_x(dasp):
    rp, ci, v = util.mixed_matrix(500, 1000, 21)
    bounds = np.array([0, 300, 650, 1000], np.int32)
    stride = 384
    plan = dasp.Plan(rp, ci, v, 1000, part_bounds=bounds, part_stride=stride)
    rows = util.decode_plan(plan)
    g = np.searchsorted(bounds, ci, side="right") - 1
    want = g * stride + (ci - bounds[g])
    order = plan.order_rid
    for slot in range(500):
        r = order[slot]
        assert rows[slot][0] == want[rp[r]:rp[r + 1]].tolist()
    assert plan.x_len == 3 * stride


def test_partition_rows(dasp):
    rp, _, _ = util.mixed_matrix(1000, 100, 2)
    for g in (1, 2, 3, 8):
        b = dasp.partition_rows(rp, g)
        assert b[0] == 0 and b[-1] == 1000 and (np.diff(b) >= 0).all()
        per = np.diff(rp[b])
        assert per.max() - rp[-1] / g <= np.diff(rp).max()


def test_synthetic_generators_are_sliceable_and_seeded(dasp):
    for name in dasp.SYNTH_NAMES:
        sc = 0.002 if name != "cop20k_A" else 0.05
        rows, cols = dasp.synth_dims(name, sc)
        rp, ci = dasp.synth_csr(name, sc)
        assert rp.size == rows + 1 and (ci >= 0).all() and (ci < cols).all()
        a, b = rows // 3, 2 * rows // 3
        rp2, ci2 = dasp.synth_csr(name, sc, a, b)
        assert (ci2 == ci[rp[a]:rp[b]]).all() and (np.diff(rp2) == np.diff(rp)[a:b]).all()
    # the symmetric stand-ins really are symmetric
    import scipy.sparse as sp
    for name, sc in (("cop20k_A", 0.05), ("nlpkkt160", 0.001), ("Queen_4147", 0.002)):
        rows, cols = dasp.synth_dims(name, sc)
        rp, ci = dasp.synth_csr(name, sc)
        A = sp.csr_matrix((np.ones(ci.size), ci, rp), shape=(rows, cols))
        assert (A != A.T).nnz == 0, name


def banded_matrix(m, band, seed, mean_len=20):
    rng = np.random.default_rng(seed)
    lens = np.clip(rng.normal(mean_len, mean_len * 0.6, m), 0, 4 * mean_len).astype(np.int64)
    rp = np.zeros(m + 1, np.int64)
    np.cumsum(lens, out=rp[1:])
    rows = np.repeat(np.arange(m), lens)
    ci = np.clip(rows + rng.integers(-band, band + 1, rows.size), 0, m - 1).astype(np.int32)
    return rp.astype(np.int32), ci, rng.uniform(-1, 1, rows.size)


@pytest.mark.parametrize("prec", [64, 16])
@pytest.mark.parametrize("y_order", [0, 1])
def test_windowed_mode_keeps_reference_slots_and_decodes(dasp, oracle, prec, y_order):
    dt = np.float64 if prec == 64 else np.float16
    rp, ci, v = banded_matrix(5000, 1500, 3)
    v = v.astype(dt)
    plan = dasp.Plan(rp, ci, v, 5000, precision=prec, y_order=y_order)          # auto: narrow band -> windows on
    st = plan.stats
    assert st["x_window_on"] == 1 and st["n_windows_lds"] == st["n_windows"] > 0 and st["lds_bytes"] <= 81920
    assert st["window_nnz_frac"] == 1.0
    P = oracle.Packed(prec, rp, ci, v.astype(np.float64), 5000)
    assert (plan.order_rid == P.order_rid).all()                                  # output permutation unchanged
    rows = util.decode_plan(plan)
    order = plan.order_rid
    assert sorted(rows) == list(range(5000))
    for slot in range(5000):
        r = order[slot]
        assert rows[slot][0] == ci[rp[r]:rp[r + 1]].tolist()
    # every window's LDS span covers all columns of its rows
    cmin, wlen = plan.host_array("win_cmin"), plan.host_array("win_len")
    dst = plan.host_array("med_dst")
    R = st["row_window"]
    for w in range(st["n_windows"]):
        for pos in range(w * R, min((w + 1) * R, dst.size)):        # the MFMA-block rows (shorter medium rows are slabs)
            r = dst[pos] if y_order == 1 else order[dst[pos]]
            cols = ci[rp[r]:rp[r + 1]]
            assert cols.min() >= cmin[w] and cols.max() < cmin[w] + wlen[w]


def test_windowed_mode_auto_off_and_forced(dasp):
    rp, ci, v = util.mixed_matrix(3000, 2_000_000, 7)          # columns all over a wide matrix: windows do not fit
    assert dasp.Plan(rp, ci, v, 2_000_000).stats["x_window_on"] == 0
    assert dasp.Plan(rp, ci, v, 2_000_000, x_window=-1).stats["x_window_on"] == 0
    rp, ci, v = banded_matrix(3000, 200, 5)
    assert dasp.Plan(rp, ci, v, 3000, x_window=-1).stats["x_window_on"] == 0
    st = dasp.Plan(rp, ci, v, 3000, x_window=4096, row_window=64).stats            # tiny cap: only some windows fit
    assert st["x_window_on"] == 1 and st["row_window"] == 64 and st["lds_bytes"] <= 4096


@pytest.mark.parametrize("prec", [64, 16])
def test_cid16_mode_decodes_and_auto_rule(dasp, prec):
    dt = np.float64 if prec == 64 else np.float16
    # narrow columns: every chunk spans < 65535; auto keeps 32-bit ids on a matrix this small (cache resident)
    rp, ci, v = banded_matrix(4000, 900, 12)
    assert dasp.Plan(rp, ci, v.astype(dt), 4000, precision=prec, x_window=-1).stats["cid16_on"] == 0
    plan = dasp.Plan(rp, ci, v.astype(dt), 4000, precision=prec, x_window=-1, cid16=1)
    assert plan.stats["cid16_on"] == 1 and plan.host_array("med_cid").size == 0
    rows = util.decode_plan(plan)
    order = plan.order_rid
    for slot in range(4000):
        r = order[slot]
        assert rows[slot][0] == ci[rp[r]:rp[r + 1]].tolist()
    # columns all over a 2M-wide matrix: chunks do not compress -> auto off, forcing moves them to the tails
    rp, ci, v = util.mixed_matrix(3000, 2_000_000, 7)
    off = dasp.Plan(rp, ci, v.astype(dt), 2_000_000, precision=prec)
    assert off.stats["cid16_on"] == 0
    forced = dasp.Plan(rp, ci, v.astype(dt), 2_000_000, precision=prec, cid16=1)
    assert forced.stats["cid16_on"] == 1 and forced.stats["nnz_irreg"] > off.stats["nnz_irreg"]
    rows = util.decode_plan(forced)
    order = forced.order_rid
    for slot in range(3000):
        r = order[slot]
        assert rows[slot][0] == ci[rp[r]:rp[r + 1]].tolist()
    assert dasp.Plan(rp, ci, v.astype(dt), 2_000_000, precision=prec, cid16=-1).stats["cid16_on"] == 0


def test_cid16_span_boundary(dasp):
    """a chunk spanning exactly 65534 columns compresses, 65535 does not"""
    for span, want in ((65534, 1), (65535, 0)):
        lens = [8] * 16
        rp = np.arange(0, 8 * 17, 8, dtype=np.int32)
        ci = np.tile(np.array([0, 1, 2, span, 70000, 70001, 70002, 70003], np.int32), 16)
        plan = dasp.Plan(rp, ci, np.ones(ci.size), 80000, x_window=-1, cid16=1, slab_max_len=4)     # rows of 8 as MFMA blocks
        st = plan.stats
        assert (st["nnz_irreg"] == 0) == bool(want), span      # the wide chunk ends the regular part: its rows go to the tail
        rows = util.decode_plan(plan)
        for slot in range(16):
            assert rows[slot][0] == ci[:8].tolist()


@pytest.mark.parametrize("prec", [64, 16])
def test_chunk_pairs_layout(dasp, prec):
    """paired medium chunks (plan.hpp med_npair / med_elem_index): which chunks of a block are stored [pair][lane][2][vpl], for every
    mode of the option; the decoder un-pairs them, so rows, values and order_rid are those of the unpaired plan"""
    dt = np.float64 if prec == 64 else np.float16
    K, CH, vpl = (4, 64, 1) if prec == 64 else (16, 256, 4)
    batch, shot = (4, 8) if prec == 64 else (2, 2)
    # 16 rows of 11 K (pipelined: 11 chunks), 16 rows of 3 K (one shot, no tail), 16 rows of 3 K + 1 (one shot + a tail step)
    lens = np.array([11 * K] * 16 + [3 * K + 1] * 16 + [3 * K] * 16)
    rp, ci, v = util.csr_from_lengths(lens, 5000, 3, dtype=dt)
    ref = None
    for mode, want in ((-1, (0, 0, 0)), (1, (11 // batch * batch, 2 if prec == 16 else 0, 2 if prec == 16 else 0)),
                       (2, (11 // batch * batch, 2 if prec == 16 else 0, 2))):
        plan = dasp.Plan(rp, ci, v, 5000, precision=prec, x_window=-1, slab_max_len=4, chunk_pairs=mode, piece_min_len=-1)
        st = plan.stats
        assert st["chunk_pairs"] == max(mode, 0) and plan.host_array("med_ptr").tolist() == [0, 11, 14, 17]
        mv, ip_ = plan.host_array("med_val"), plan.host_array("irr_ptr")
        assert (np.diff(ip_)[16:32] == 1).all() and ip_[16] == 0 and ip_[-1] == 16
        for b, npair in enumerate(want):
            c0 = [0, 11, 14][b]
            blk = mv[c0 * CH:]
            row = 16 * b                                  # the block's first row, its entries 0 .. K-1 of chunk 0 and chunk 1
            r = plan.order_rid[row]
            a = v[rp[r]:rp[r + 1]]
            lane = [k * 16 for k in range(K)] if prec == 64 else [(k // 4) * 16 for k in range(K)]
            j = [0] * K if prec == 64 else [k % 4 for k in range(K)]
            for c in (0, 1):
                got = [blk[(c & ~1) * CH + vpl * (2 * lane[k] + (c & 1)) + j[k]] if c < npair else blk[c * CH + vpl * lane[k] + j[k]] for k in range(K)]
                assert got == a[c * K:(c + 1) * K].tolist(), (mode, b, c)
        rows = util.decode_plan(plan)
        if ref is None:
            ref = rows
        assert rows == ref and all(rows[s][0] == ci[rp[plan.order_rid[s]]:rp[plan.order_rid[s] + 1]].tolist() for s in range(48))
    # automatic: mode 1 below 1 GiB of CSR, nothing paired in a windowed plan
    assert dasp.Plan(rp, ci, v, 5000, precision=prec, x_window=-1).stats["chunk_pairs"] == 1
    w = dasp.Plan(rp, ci, v, 5000, precision=prec, x_window=100000, chunk_pairs=2).stats
    assert w["x_window_on"] == 1 and w["chunk_pairs"] == 0


def test_cid8_narrow_chunks_layout(dasp, tmp_path):
    """one-byte ids (f64, 16-bit-id plans, pipelined blocks): chunks whose columns span <= 254 go to the front of the block's paired region in
    whole batches of four, ids [batch][lane][4]; values, base and ids move together (med_korig); the decoder, the plan file and cid8 = -1"""
    K, CH = 4, 64
    # 16 identical rows of 12 chunks: chunks 0..3 wide (columns 1000 apart), 4..9 narrow (adjacent columns), 10 spans exactly 254, 11 spans 255
    cols = []
    for c in range(12):
        lo = 100000 * c
        cols += [lo, lo + 1000, lo + 2000, lo + 3000] if c < 4 else [lo, lo + 1, lo + 2, lo + (254 if c == 10 else 255 if c == 11 else 3)]
    ci = np.tile(np.array(cols, np.int32), 16)
    rp = np.arange(0, 48 * 17, 48, dtype=np.int32)
    v = np.arange(1, ci.size + 1, dtype=np.float64)
    plan = dasp.Plan(rp, ci, v, 1300000, x_window=-1, cid16=1, slab_max_len=4, piece_min_len=-1)
    st = plan.stats
    assert st["cid16_on"] == 1 and st["chunk_pairs"] == 1 and plan.host_array("med_ptr").tolist() == [0, 12]
    # 7 narrow chunks (4..10) -> one batch of four in front; the paired region is all 12 chunks
    assert st["cid8_chunks"] == 4 and plan.host_array("med_c8ptr").tolist() == [0, 4]
    assert plan.host_array("med_korig").tolist() == [4, 5, 6, 7, 0, 1, 2, 3, 8, 9, 10, 11]
    assert plan.host_array("med_base").tolist() == [100000 * k for k in (4, 5, 6, 7, 0, 1, 2, 3, 8, 9, 10, 11)]
    c8 = plan.host_array("med_cid8").reshape(64, 4)              # [lane][chunk of the batch]; lane = k * 16 + row
    assert plan.host_array("med_cid8").size == 4 * CH and plan.host_array("med_cid16").size == 8 * CH
    for q in range(4):
        assert c8[:, q].reshape(4, 16)[:, 3].tolist() == [0, 1, 2, 3]
    rows = util.decode_plan(plan)
    order = plan.order_rid
    for slot in range(16):
        r = order[slot]
        assert rows[slot][0] == cols and rows[slot][1] == v[rp[r]:rp[r + 1]].tolist()
    f = str(tmp_path / "p.plan")
    plan.save(f)
    back = dasp.Plan.load(f)
    assert util.decode_plan(back) == rows and back.stats["cid8_chunks"] == 4
    off = dasp.Plan(rp, ci, v, 1300000, x_window=-1, cid16=1, slab_max_len=4, piece_min_len=-1, cid8=-1)
    assert off.stats["cid8_chunks"] == 0 and off.host_array("med_cid8").size == 0 and off.host_array("med_korig").tolist() == list(range(12))
    assert util.decode_plan(off) == rows
    # one-shot blocks (<= 8 steps) keep 16-bit ids unless they are paired as a whole (chunk_pairs = 2: plans far beyond the Infinity Cache; r4),
    # f16 plans always
    rp8, ci8 = np.arange(0, 32 * 17, 32, dtype=np.int32), np.tile(np.array(cols[16:48], np.int32), 16)      # 8 chunks: the narrow 4..9, then 10 (254) and 11 (255)
    short = dasp.Plan(rp8, ci8, np.ones(32 * 16), 1300000, x_window=-1, cid16=1, slab_max_len=4, piece_min_len=-1)
    assert short.stats["cid8_chunks"] == 0 and short.stats["chunk_pairs"] == 1
    v8 = np.arange(1, ci8.size + 1, dtype=np.float64)
    assert dasp.Plan(rp8, ci8, v8, 1300000, x_window=-1, cid16=1, slab_max_len=4, piece_min_len=-1, chunk_pairs=2).stats["cid8_chunks"] == 0      # late r5: automatic keeps one-byte ids out of one-shot blocks
    one = dasp.Plan(rp8, ci8, v8, 1300000, x_window=-1, cid16=1, slab_max_len=4, piece_min_len=-1, chunk_pairs=2, cid8=1)
    # 7 of its 8 chunks are narrow -> three whole PAIRS in front; ids [pair][lane][2 bytes]
    assert one.stats["chunk_pairs"] == 2 and one.stats["cid8_chunks"] == 6 and one.host_array("med_c8ptr").tolist() == [0, 6]
    assert one.host_array("med_korig").tolist() == [0, 1, 2, 3, 4, 5, 6, 7] and one.host_array("med_cid8").size == 6 * CH and one.host_array("med_cid16").size == 2 * CH
    c8 = one.host_array("med_cid8").reshape(3, 64, 2)             # [pair][lane][chunk of the pair]; lane = k * 16 + row
    for pr in range(3):
        for h in range(2):
            assert c8[pr, :, h].reshape(4, 16)[:, 5].tolist() == [0, 1, 2, 254 if 2 * pr + h == 6 else 3]
    rows8 = util.decode_plan(one)
    for slot in range(16):
        r = one.order_rid[slot]
        assert rows8[slot][0] == cols[16:48] and rows8[slot][1] == v8[rp8[r]:rp8[r + 1]].tolist()
    f8 = str(tmp_path / "p8.plan")
    one.save(f8)
    assert util.decode_plan(dasp.Plan.load(f8)) == rows8
    half = dasp.Plan(rp, ci, np.ones(ci.size, np.float16), 1300000, precision=16, x_window=-1, cid16=1, slab_max_len=4, piece_min_len=-1)
    assert half.stats["cid8_chunks"] == 0


@pytest.mark.parametrize("prec", [64, 16])
def test_serialised_plan_round_trip(dasp, tmp_path, prec):
    dt = np.float64 if prec == 64 else np.float16
    rp, ci, v = banded_matrix(3000, 400, 4)
    for kw in (dict(), dict(x_window=-1, cid16=1, y_order=1), dict(part_bounds=np.array([0, 1000, 3000], np.int32), part_stride=2048)):
        plan = dasp.Plan(rp, ci, v.astype(dt), 3000, precision=prec, **kw)
        path = str(tmp_path / "p.daspplan")
        plan.save(path)
        back = dasp.Plan.load(path)
        assert back.stats == plan.stats and (back.order_rid == plan.order_rid).all()
        assert (back.y_order, back.x_len, back.precision) == (plan.y_order, plan.x_len, prec)
        for name in ("long_val long_cid piece_ptr piece_dst multi_ptr multi_dst med_ptr med_val med_cid med_cid16 med_cid8 med_c8ptr med_korig med_base "
                     "irr_ptr irr_val irr_cid med_dst win_cmin win_len short_val short_cid short_groups").split():
            a, b = plan.host_array(name), back.host_array(name)
            assert a.dtype == b.dtype and np.array_equal(a, b), name
    with open(path, "r+b") as f:
        f.truncate(200)
    with pytest.raises(dasp.DaspError):
        dasp.Plan.load(path)


@pytest.mark.parametrize("prec", [64, 16])
def test_auto_hybrid_windows_need_even_rows_and_keep_the_format(dasp, prec):
    """rows of equal length whose columns form a band plus 10 % outliers: the strict windows cannot fit (every window spans the
    matrix), auto stages the densest span (hybrid) and the packed plan still decodes to the rows; power-law rows with the same
    columns leave it off (the window workgroups would cost more than they save)"""
    rng = np.random.default_rng(23)
    m = n = 120000
    dt = np.float64 if prec == 64 else np.float16

    def cols(rows, half=6000, near=0.95):
        return np.where(rng.random(rows.size) < near, np.clip(rows + rng.integers(-half, half + 1, rows.size), 0, n - 1), rng.integers(0, n, rows.size)).astype(np.int32)
    rp = (np.arange(m + 1, dtype=np.int64) * 12).astype(np.int32)
    # r6: a NARROW band (+-500 columns: a row's gathers span a few KB, and rows of one length keep their neighbours in a block) stays without windows -- the L1 serves it;
    # measured: 120 k such rows f64 7.5 us with hybrid windows against 5.7 without, 1 M rows 61.5 / 44.2 (plan.cpp, the core-span condition)
    narrow = cols(np.repeat(np.arange(m), 12), 500, 0.9)
    assert dasp.Plan(rp, narrow, np.ones(narrow.size, dt), n, precision=prec, slab_max_len=4).stats["x_window_on"] == 0
    ci = cols(np.repeat(np.arange(m), 12))
    plan = dasp.Plan(rp, ci, np.ones(ci.size, dt), n, precision=prec)
    st = plan.stats
    assert st["x_window_on"] == 1 and st["x_window_hybrid"] == 1 and 0.6 < st["window_nnz_frac"] < 0.97 and st["lds_bytes"] <= 81920, st
    rows = util.decode_plan(plan)
    order = plan.order_rid
    for slot in list(range(0, m, 997)):
        r = order[slot]
        assert sorted(rows[slot][0]) == sorted(ci[rp[r]:rp[r + 1]].tolist())
    assert dasp.Plan(rp, ci, np.ones(ci.size, dt), n, precision=prec, x_window_hybrid=-1).stats["x_window_on"] == 0
    lens = np.minimum(5 + (rng.pareto(1.3, m) * 3).astype(np.int64), 250)
    rp2 = np.zeros(m + 1, np.int64)
    np.cumsum(lens, out=rp2[1:])
    ci2 = cols(np.repeat(np.arange(m), lens))
    st2 = dasp.Plan(rp2.astype(np.int32), ci2, np.ones(ci2.size, dt), n, precision=prec).stats
    assert st2["x_window_hybrid"] == 0


@pytest.mark.parametrize("prec", [64, 16])
def test_densest_span_histogram_search_equals_the_sort_search(dasp, prec, monkeypatch):
    """hybrid windows: the densest LDS-sized span of a window's columns is found by a histogram over the 16-byte-aligned column groups
    and a sliding sum where the window's columns are dense enough (r3: it replaced a sort of every window's columns, 0.43 s of
    nlpkkt160's preprocessing), by the sort otherwise -- both must pick the same span (first start with the largest count)"""
    rng = np.random.default_rng(29)
    m = n = 150000
    dt = np.float64 if prec == 64 else np.float16
    rows = np.repeat(np.arange(m), 14)
    near = rng.random(rows.size) < 0.8
    ci = np.clip(rows + np.where(near, rng.integers(-2500, 2501, rows.size), rng.integers(-14000, 14001, rows.size)), 0, n - 1).astype(np.int32)
    rp = (np.arange(m + 1, dtype=np.int64) * 14).astype(np.int32)
    out = {}
    for mode in ("hist", "sort"):
        if mode == "sort":
            monkeypatch.setenv("DASP_HYBRID_SORT", "1")
        plan = dasp.Plan(rp, ci, np.ones(ci.size, dt), n, precision=prec, x_window_hybrid=1)
        st = plan.stats
        assert st["x_window_on"] == 1 and st["x_window_hybrid"] == 1
        out[mode] = (plan.host_array("win_cmin").copy(), plan.host_array("win_len").copy(), st["window_nnz_frac"], st["lds_bytes"])
        plan.close()
    monkeypatch.delenv("DASP_HYBRID_SORT")
    np.testing.assert_array_equal(out["hist"][0], out["sort"][0])
    np.testing.assert_array_equal(out["hist"][1], out["sort"][1])
    assert out["hist"][2:] == out["sort"][2:] and 0.5 < out["hist"][2] <= 1.0


def test_medium_rows_as_pieces_keep_slots_and_counters(dasp, oracle):
    """piece_min_len: the longest medium rows stored as wave-sized pieces -- order_rid and every classifier counter are still the
    oracle's, the packed arrays decode to the rows, the rows concerned sit in the first medium slots"""
    rng = np.random.default_rng(5)
    lens = np.concatenate([rng.choice([5, 9, 17, 40], 3000), rng.choice([100, 180, 255], 60), [300, 1200], rng.choice([1, 2, 3, 4, 0], 500)])
    rng.shuffle(lens)
    rp, ci, v = util.csr_from_lengths(lens, 5000, 3)
    for prec in (64, 16):
        dt = np.float64 if prec == 64 else np.float16
        P = oracle.Packed(prec, rp, ci, v, 5000)
        for kw in (dict(piece_min_len=100, x_window=-1), dict(piece_min_len=17, y_order=1, x_window=-1), dict()):
            plan = dasp.Plan(rp, ci, v.astype(dt), 5000, precision=prec, **kw)
            st = plan.stats
            if kw.get("piece_min_len") == 100:
                assert st["med_rows_as_pieces"] == 60
            if "piece_min_len" in kw:
                assert st["med_rows_as_pieces"] > 0
            for f in "row_long row_block row_zero short_row_1 short_row_2 short_row_3 short_row_4 common_13 nnz_short nnz_long".split():
                assert st[f] == getattr(P, f), f
            assert (plan.order_rid == P.order_rid).all()
            rows = util.decode_plan(plan)
            order = plan.order_rid
            assert len(rows) == lens.size
            for slot, (cs, vs) in rows.items():
                r = order[slot]
                assert cs == ci[rp[r]:rp[r + 1]].tolist()
            plan.close()


@pytest.mark.parametrize("prec", [64, 16])
def test_sort_columns_sorts_the_rows_stably_and_only_when_asked(dasp, prec):
    """sort_columns = 1: every row whose columns do not ascend is packed with its (column, value) pairs in column order -- stable, duplicates keep their CSR order --,
    order_rid and the counters unchanged; 0: the CSR order, as ever; rows that ascend already give the same plan either way"""
    dt = np.float64 if prec == 64 else np.float16
    rp, ci, v = util.mixed_matrix(1500, 300, 5, values="f16" if prec == 16 else "uniform", dtype=dt)     # 300 columns: duplicates inside the longer rows
    assert any(len(set(ci[rp[r]:rp[r + 1]].tolist())) < rp[r + 1] - rp[r] for r in range(1500))
    for kw in (dict(), dict(y_order=1), dict(x_window=-1, cid16=1, chunk_pairs=2)):
        plain = dasp.Plan(rp, ci, v, 300, precision=prec, **kw)
        srt = dasp.Plan(rp, ci, v, 300, precision=prec, sort_columns=1, **kw)
        assert (plain.order_rid == srt.order_rid).all()
        a, b = plain.stats, srt.stats
        a.pop("pre_ms"), b.pop("pre_ms")
        for k in ("short_row_1", "row_long", "row_block", "row_zero", "nnz_short", "nnz_long", "common_13"):
            assert a[k] == b[k], k
        order = srt.order_rid
        for slot, (cs, vs) in util.decode_plan(srt).items():
            r = int(order[slot])
            pairs = sorted(zip(ci[rp[r]:rp[r + 1]].tolist(), v[rp[r]:rp[r + 1]].tolist()), key=lambda t: t[0])      # Python's sort is stable
            assert cs == [c for c, _ in pairs] and list(vs) == [x for _, x in pairs], r
        for slot, (cs, vs) in util.decode_plan(plain).items():
            r = int(plain.order_rid[slot])
            assert cs == ci[rp[r]:rp[r + 1]].tolist()
        plain.close(); srt.close()
    # ascending rows: nothing to do, the same arrays
    rowid = np.repeat(np.arange(1500), np.diff(rp))
    o = np.lexsort((ci, rowid))
    p0, p1 = dasp.Plan(rp, ci[o], v[o], 300, precision=prec), dasp.Plan(rp, ci[o], v[o], 300, precision=prec, sort_columns=1)
    for name in ("med_val", "med_cid", "med_cid16", "irr_cid", "irr_val", "short_cid", "short_val", "long_cid", "long_val"):
        assert np.array_equal(p0.host_array(name), p1.host_array(name)), name


def test_long_rows_stay_one_piece_when_none_is_very_long(dasp):
    """long_piece = 0 (default): one piece per long row -- no stage-2 launch -- when the long rows hold <= 2 M nonzeros in rows of <= 16384, or when no
    row exceeds 4096 whatever their number; beyond that, pieces of 1024 and a partial sum per piece"""
    def multi(lens, n_cols):
        rp, ci, v = util.csr_from_lengths(np.asarray(lens), n_cols, 3, values="ones")
        p = dasp.Plan(rp, ci, v, n_cols)
        st = p.stats
        p.close()
        return st["n_long_multi"], st["n_long_pieces"], st["row_long"]
    assert multi([3000] * 20 + [7] * 100, 6000) == (0, 20, 20)                    # few long nonzeros
    assert multi([4000] * 600 + [7] * 100, 8000) == (0, 600, 600)                 # 2.4 M long nonzeros, but no row beyond 4096
    m, pieces, rl = multi([5000] * 450 + [7] * 100, 8000)                         # 2.25 M long nonzeros in rows of 5000: pieces of 1024
    assert m == 450 and pieces == 450 * 5 and rl == 450
    assert multi([20000] * 3 + [7] * 100, 30000)[0] == 3                          # rows beyond 16384 are always cut
    # r5: ... and a row is cut when walking it alone (one wave, ~0.35 us per batch of 256 f64 / 512 f16 elements) would outlast everything else of the launch
    # plus the stage-2 launch that one piece per row saves (~7 us + the matrix at 5 TB/s)
    assert multi([15000] * 3 + [7] * 1000, 30000) == (3, 45, 3)                   # 20 us of chain in a ~4 us launch: pieces of 1024
    assert multi([4500] * 3 + [7] * 1000, 30000) == (0, 3, 3)                     # 6 us of chain: not worth a second launch
    assert multi([10000] + [10] * 600000, 20000) == (0, 1, 1)                     # 13.7 us of chain beside 6 M nonzeros (~21 us): stays whole
    assert multi([10000] + [10] * 100000, 20000) == (1, 10, 1)                    # ... beside 1 M nonzeros (~9.4 us): cut


def test_corrupt_plan_files_are_rejected_not_trusted(dasp, tmp_path):
    """dasp_plan_load re-derives what the kernels index with: a plan file with one damaged table / column id / header field comes
    back as an error (never an exception through the C ABI, never a plan that would read out of bounds on upload)"""
    rp, ci, v = util.mixed_matrix(2500, 2000, 13)
    good = str(tmp_path / "g.plan")
    for kw in (dict(), dict(x_window=100000, row_window=128), dict(cid16=1), dict(col_panels=3), dict(col_panels=2, row_tile_max=8),
               dict(precision=16, two_phase=1, tp_col_block=256, tp_row_block=64)):
        dasp.Plan(rp, ci, v.astype(np.float16) if kw.get("precision") == 16 else v, 2000, **kw).save(good)
        blob = bytearray(open(good, "rb").read())
        dasp.Plan.load(good).close()                                   # the undamaged file loads
        rng = np.random.default_rng(1)
        rejected = 0
        for trial in range(60):
            bad = bytearray(blob)
            at = int(rng.integers(8, len(bad) - 4))
            if trial % 3 == 0:                                         # a wild 32-bit value (ids, pointers, counts)
                bad[at:at + 4] = int(rng.integers(1 << 28, 1 << 31)).to_bytes(4, "little")
            elif trial % 3 == 1:                                       # a negative one
                bad[at:at + 4] = (-int(rng.integers(2, 1 << 30))).to_bytes(4, "little", signed=True)
            else:                                                      # a length field claiming more than the file holds
                bad[at:at + 8] = (1 << 40).to_bytes(8, "little")
            q = str(tmp_path / "b.plan")
            open(q, "wb").write(bad)
            try:
                dasp.Plan.load(q).close()                              # a hit inside a value array is harmless and loads
            except dasp.DaspError as e:
                assert e.status in (-2, -5, -11)
                rejected += 1
        assert rejected >= 5
    for old in (b"4", b"5", b"6"):                                     # older layouts' magics ('5': before the row tiles / 19-int header; '6': before the two-phase streams / 22-int header)
        bad = bytearray(blob)
        assert bytes(bad[:8]) == b"DASPPLN8"
        bad[7:8] = old
        open(good, "wb").write(bad)
        with pytest.raises(dasp.DaspError) as e:
            dasp.Plan.load(good)
        assert e.value.status == -2


def test_plan_files_with_two_writers_for_one_y_are_rejected(dasp, tmp_path):
    """put_y assumes ONE writer per y index (a plain store; a plain read-modify-write in accumulate mode): a file whose destination
    tables name a row twice -- or stage hybrid windows without windows -- is refused on load"""
    rp, ci, v = util.mixed_matrix(2500, 2000, 13)
    path = str(tmp_path / "p.plan")

    def damaged(plan, name, mutate, holder=None):
        plan.save(path)
        blob = bytearray(open(path, "rb").read())
        arr = (holder or plan).host_array(name)
        raw = arr.tobytes()
        key = len(raw).to_bytes(8, "little") + raw                       # every array is stored behind its int64 byte count
        at = bytes(blob).find(key)
        assert at > 0 and bytes(blob).find(key, at + 1) < 0              # located unambiguously
        at += 8
        new = arr.copy()
        mutate(new)
        blob[at:at + len(raw)] = new.tobytes()
        open(path, "wb").write(blob)
        with pytest.raises(dasp.DaspError) as e:
            dasp.Plan.load(path)
        assert e.value.status in (-2, -5) and "writer" in str(e.value)

    def dup(a):
        a[1] = a[0]

    plan = dasp.Plan(rp, ci, v, 2000, y_order=dasp.Y_NATURAL)            # natural order: destinations go through order_rid
    damaged(plan, "order", dup)
    plan.close()
    plan = dasp.Plan(rp, ci, v, 2000, x_window=100000, row_window=128)   # windowed: med_dst
    damaged(plan, "med_dst", dup)
    plan.close()
    lens = np.array([300, 400, 500] + [7] * 200)
    rp2, ci2, v2 = util.csr_from_lengths(lens, 900, 3)
    plan = dasp.Plan(rp2, ci2, v2, 900, y_order=dasp.Y_NATURAL)           # long rows: piece_dst
    damaged(plan, "piece_dst", dup)
    plan.close()
    plan = dasp.Plan(rp, ci, v, 2000, col_panels=2, row_tile_max=4)       # a panel's row tiles claiming the positions of its blocks' rows too
    sub = plan.panel(0)[0]
    assert sub.stats["row_tile_nnz"] > 0 and sub.stats["row_block"] > 0

    def all_on(a):
        a[:-1] = np.uint64(0xFFFFFFFFFFFFFFFF)

    damaged(plan, "rt_mask", all_on, holder=sub)
    plan.close()


def test_mg_plan_splits_a_row_slice_by_column_owner(dasp):
    """dasp_mg_plan_create (multigpu.cpp, host part: no GPU needed): the own-column / other-column plans behind the overlapped
    all-gather hold every entry of the rank's rows exactly once -- own columns re-based to the rank's slice, the others remapped
    to the padded all-gather layout -- for every rank of a 3-way partition, with and without the split."""
    from dasp_amd.multi import MgPlan
    m = n = 900
    rp, ci, v = util.mixed_matrix(m, n, 31)
    world = 3
    bounds = dasp.partition_rows(rp, world)
    for overlap in (True, False):
        for rank in range(world):
            r0, r1 = int(bounds[rank]), int(bounds[rank + 1])
            sl = slice(rp[r0], rp[r1])
            mg = MgPlan(rp[r0:r1 + 1] - rp[r0], ci[sl], v[sl], m, n, bounds, rank, overlap=overlap)
            info = mg.info
            stride = info["stride"]
            assert stride % 64 == 0 and stride >= np.diff(bounds).max() and info["overlap"] == int(overlap)
            assert info["nnz_own"] + info["nnz_other"] == rp[r1] - rp[r0]
            own, oth = mg.subplan(0), mg.subplan(1)
            assert (oth is not None) == (overlap and info["nnz_other"] > 0)
            rows_own = util.decode_plan(own)
            rows_oth = util.decode_plan(oth) if oth is not None else {}
            o_own = own.order_rid
            o_oth = oth.order_rid if oth is not None else None
            by_row = {}
            for slot, (cs, vs) in rows_own.items():
                cs = np.asarray(cs, np.int64)
                glob = cs + r0 if overlap else bounds[cs // stride] + cs % stride       # own slice / gathered layout -> global column
                by_row.setdefault(int(o_own[slot]), []).extend(zip(glob.tolist(), vs))
            for slot, (cs, vs) in rows_oth.items():
                cs = np.asarray(cs, np.int64)
                glob = bounds[cs // stride] + cs % stride
                assert not ((glob >= r0) & (glob < r1)).any()
                by_row.setdefault(int(o_oth[slot]), []).extend(zip(glob.tolist(), vs))
            for r in range(r1 - r0):
                want = sorted(zip(ci[rp[r0 + r]:rp[r0 + r + 1]].tolist(), v[rp[r0 + r]:rp[r0 + r + 1]].tolist()))
                assert sorted(by_row.get(r, [])) == want
            mg.close()


def test_mg_plan_argument_errors(dasp):
    from dasp_amd.multi import MgPlan
    rp, ci, v = util.mixed_matrix(100, 100, 3)
    with pytest.raises(dasp.DaspError):
        MgPlan(rp, ci, v, 100, 100, np.array([0, 50, 90], np.int32), 0)          # bounds do not reach rowA
    with pytest.raises(dasp.DaspError):
        MgPlan(rp, ci, v, 100, 100, np.array([0, 100], np.int32), 1)             # rank out of range
    with pytest.raises(dasp.DaspError):
        MgPlan(rp, ci + 1000, v, 100, 100, np.array([0, 100], np.int32), 0)      # column out of range
    bad = rp.copy(); bad[40] = bad[41] + 5
    with pytest.raises(dasp.DaspError):
        MgPlan(bad, ci, v, 100, 100, np.array([0, 100], np.int32), 0)             # row pointer not monotone
    mg = MgPlan(rp, ci, v, 100, 100, np.array([0, 100], np.int32), 0)
    with pytest.raises(dasp.DaspError) as e:
        mg.spmv()                                                                # not uploaded
    assert e.value.status == -22


def test_auto_windows_skip_rows_with_adjacent_columns(dasp):
    """FEM-like rows gather runs of neighbouring columns: even when their windows fit in LDS, auto leaves staging off"""
    for name, scale in (("HV15R", 0.03), ("nlpkkt160", 0.01)):
        rp, ci = dasp.synth_csr(name, scale)
        _, n = dasp.synth_dims(name, scale)
        v = np.ones(ci.size)
        assert dasp.Plan(rp, ci, v, n).stats["x_window_on"] == 0
        st = dasp.Plan(rp, ci, v, n, x_window=81920).stats                       # they do fit: forcing turns it on
        assert st["x_window_on"] == 1 and st["n_windows_lds"] == st["n_windows"] > 0


@pytest.mark.parametrize("prec", [64, 16])
def test_slab_layout_decodes_and_keeps_the_reference_slots(dasp, oracle, prec):
    """medium rows stored as slabs: order_rid and the classifier counters are the oracle's, the packed arrays decode to the rows"""
    dt = np.float64 if prec == 64 else np.float16
    lens = np.random.default_rng(12).choice([0, 1, 3, 5, 5, 6, 7, 9, 14, 17, 25, 33, 70, 300], size=2500)
    rp, ci, v = util.csr_from_lengths(lens, 4000, 5, dtype=dt)
    for smax in (4, 7, 16, 32):
        plan = dasp.Plan(rp, ci, v, 4000, precision=prec, slab_max_len=smax, x_window=-1)
        P = oracle.Packed(prec, rp, ci, v.astype(np.float64), 4000)
        assert (plan.order_rid == P.order_rid).all() and plan.stats["row_block"] == P.row_block
        n_slab = int(((lens >= 5) & (lens <= smax)).sum())
        assert plan.host_array("irr_ptr").size - 1 == P.row_block - n_slab
        rows = util.decode_plan(plan)
        order = plan.order_rid
        assert sorted(rows) == list(range(2500))
        for slot in range(2500):
            r = order[slot]
            assert rows[slot][0] == ci[rp[r]:rp[r + 1]].tolist()


def test_many_plans_back_to_back_on_the_worker_pool(dasp):
    """The persistent worker pool of the host packers (plan.cpp Pool) under its worst pattern: hundreds of plans one right after the other,
    loops of very different part counts following each other within microseconds, more threads than cores.  (A worker that woke late used to
    be able to take an index from the NEXT job's counter on the strength of the previous job's bounds: a part ran twice and the job was
    counted complete early -- a rare hang.)  Every plan must come out identical to the single-threaded build."""
    rng = np.random.default_rng(99)
    cases = []
    for rows in (70000, 150000, 40000, 300000):
        lens = rng.integers(0, 12, rows)
        rp = np.zeros(rows + 1, np.int32); np.cumsum(lens, out=rp[1:])
        ci = rng.integers(0, rows, int(rp[-1])).astype(np.int32)
        v = np.ones(int(rp[-1]))
        ref = dasp.Plan(rp, ci, v, rows, host_threads=1)
        cases.append((rp, ci, v, rows, ref.order_rid.copy(), dict(ref.stats)))
        ref.close()
    for it in range(int(os.environ.get("DASP_POOL_STRESS_ITERS", "60"))):
        rp, ci, v, rows, order, stats = cases[it % len(cases)]
        p = dasp.Plan(rp, ci, v, rows, host_threads=24)
        assert np.array_equal(p.order_rid, order)
        st = p.stats
        assert all(st[k] == stats[k] for k in ("row_long", "row_block", "n_med_blocks", "nnz_irreg", "n_short_tiles", "data_X"))
        p.close()


def test_f16_rows_without_a_regular_chunk_become_slabs_when_neighbours_are_close(dasp):
    """late r5: an f16 tile holds 16 columns of a row, so rows of fewer than 12 nonzeros never fill a regular chunk to 75 % -- their MFMA blocks are all tail steps (8 M such rows:
    0.21 of the roofline as blocks, 0.62 as slabs).  For them the automatic slab rule accepts equally long neighbours whose columns lie within 512 (f64 and rows of 12..24: within a
    128-byte line); scattered (graph-like) rows keep their blocks either way."""
    rng = np.random.default_rng(3)
    m = n = 200000
    lens = rng.integers(5, 9, m)
    rp = np.zeros(m + 1, np.int64); np.cumsum(lens, out=rp[1:])
    rows = np.repeat(np.arange(m), lens)
    k = np.arange(rp[-1]) - rp[rows]
    local = np.clip(rows + rng.integers(-256, 257, m)[rows] + k, 0, n - 1).astype(np.int32)       # a row's columns consecutive from a start within +-256 of the row
    far = rng.integers(0, n, int(rp[-1])).astype(np.int32)
    rp = rp.astype(np.int32)
    as16 = lambda ci, **kw: dasp.Plan(rp, ci, np.ones(ci.size, np.float16), n, precision=16, x_window=-1, **kw).stats
    as64 = lambda ci: dasp.Plan(rp, ci, np.ones(ci.size), n, precision=64, x_window=-1).stats
    assert as16(local)["n_med_blocks"] == 0 and as16(local)["n_short_tiles"] > 0            # slabs
    assert as16(local, slab_max_len=4)["n_med_blocks"] == (m + 15) // 16                    # (what they would have been)
    assert as16(far)["n_med_blocks"] == (m + 15) // 16                                      # scattered: blocks
    assert as64(local)["n_med_blocks"] == (m + 15) // 16                                    # f64: its tiles hold 4 columns, rows of 5..8 fill them


def test_windows_when_the_length_sort_scatters_local_rows(dasp):
    """late r5: rows of many different lengths with local columns -- the global length sort puts 16 rows from anywhere into a block.  From 16 M nonzeros on such a plan gets the
    LDS windows (sorted inside 1024 rows) although every row's own columns form runs; rows of ONE length (which keep their row order in the sort) do not"""
    rng = np.random.default_rng(4)
    m = n = 300000

    def local(lens):
        rp = np.zeros(m + 1, np.int64); np.cumsum(lens, out=rp[1:])
        rows = np.repeat(np.arange(m, dtype=np.int64), lens)
        k = np.arange(int(rp[-1]), dtype=np.int64) - rp[rows]
        start = np.clip(rows + rng.integers(-500, 501, m)[rows] - lens[rows] // 2, 0, n - lens[rows])
        return rp.astype(np.int32), (start + k).astype(np.int32)
    rp, ci = local(rng.integers(20, 120, m))                       # ~21 M nonzeros
    st = dasp.Plan(rp, ci, np.ones(ci.size), n, precision=64).stats
    assert ci.size >= (16 << 20) and st["x_window_on"] == 1 and st["n_windows_lds"] == st["n_windows"] > 0, st
    rp1, ci1 = local(np.full(m, 70))                               # 21 M nonzeros, one length: the sorted order is the row order
    assert dasp.Plan(rp1, ci1, np.ones(ci1.size), n, precision=64).stats["x_window_on"] == 0
    small = rng.integers(20, 120, m) // 4 + 5                      # the same structure under 16 M nonzeros
    rp2, ci2 = local(small)
    assert ci2.size < (16 << 20) and dasp.Plan(rp2, ci2, np.ones(ci2.size), n, precision=64).stats["x_window_on"] == 0


@pytest.mark.parametrize("prec", [64, 16])
def test_long_pieces_carry_16_bit_ids_where_their_chunks_are_narrow(dasp, prec, tmp_path):
    """r6 (reference long path src/dasp_f64.h:90-144): per chunk of a long piece the smallest column is its base; a piece all of whose chunks span
    <= 65534 columns stores u16 offsets besides the 32-bit ids (the kernel streams 10 / 4 instead of 12 / 6 bytes per nonzero), any other piece zeros.  Same lanes, same
    order of additions: order_rid, the counters and the 32-bit arrays are exactly what they were.  util.decode_plan rebuilds the narrow pieces' columns from base + offset."""
    rng = np.random.default_rng(3)
    n = 400000
    lens = [2000, 300, 5000, 257, 1500, 0, 3] + [9] * 200
    rp = np.zeros(len(lens) + 1, np.int64); np.cumsum(lens, out=rp[1:])
    cols = []
    for i, L in enumerate(lens):
        if i in (0, 3):      # local rows: consecutive columns -> narrow
            cols.append(np.arange(L) + 1000 * i)
        elif i == 2:         # sorted, dense enough per chunk: 5000 over 400 000 -> ~5000 columns per 64 entries -> narrow in f64, ~20 000 per 256 in f16 -> narrow too
            cols.append(np.sort(rng.choice(n, L, replace=False)))
        else:                # uniform, unsorted: a chunk spans most of x -> wide
            cols.append(rng.integers(0, n, L))
    ci = np.concatenate(cols).astype(np.int32)
    dt = np.float64 if prec == 64 else np.float16
    v = rng.integers(1, 9, ci.size).astype(dt)
    plan = dasp.Plan(rp.astype(np.int32), ci, v, n, precision=prec, long_piece=1024)
    pc = plan.host_array("piece_c16").reshape(-1, 2)
    pp = plan.host_array("piece_ptr")
    assert pc.shape[0] == plan.stats["n_long_pieces"] > 5 and 0 < pc[:, 1].sum() < pc.shape[0]          # both kinds present
    rows = util.decode_plan(plan)                                                                      # (asserts base = chunk minimum, narrow <=> spans fit, offsets)
    order = plan.order_rid
    for slot, (cs, vs) in rows.items():
        r = order[slot]
        assert cs == ci[rp[r]:rp[r + 1]].tolist()
    path = str(tmp_path / "l16.plan")
    plan.save(path)
    again = dasp.Plan.load(path)
    for name in ("long_cid16", "long_base", "piece_c16", "long_cid"):
        assert np.array_equal(again.host_array(name), plan.host_array(name)), name
    again.close()
    # a narrow piece whose offsets disagree with its 32-bit ids is refused
    raw = bytearray(open(path, "rb").read())
    l16 = plan.host_array("long_cid16")
    k = int(pp[int(np.flatnonzero(pc[:, 1] == 1)[0])])
    at = bytes(raw).find(l16.tobytes())
    assert at > 0
    raw[at + 2 * k: at + 2 * k + 2] = int((int(l16[k]) + 1) & 0x7FFF).to_bytes(2, "little")
    (tmp_path / "bad.plan").write_bytes(bytes(raw))
    with pytest.raises(dasp.DaspError) as e:
        dasp.Plan.load(str(tmp_path / "bad.plan"))
    assert "long_cid16" in str(e.value)
    plan.close()

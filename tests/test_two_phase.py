"""The two-phase (gather-free) form of an f16 plan (opt.two_phase; plan.hpp struct TwoPhase, twophase.cpp) on the host: the tile streams
decode back to the CSR -- rows in the plan's output order, every row's entries in CSR order --, the classifier counters and order_rid stay those
of the whole matrix, block sizes / edge cases / plan files / the validator.  The GPU side (parity with the oracle) is tests/test_gpu_spmv.py."""
import numpy as np
import pytest

import util


def _check_decodes(plan, rp, ci, v, m, natural):
    rows = util.decode_plan(plan)
    order = plan.order_rid
    lens = np.diff(rp)
    # the hybrid (r6): hub rows live column-blocked in the plan's lcb arrays, in no tile stream -- every entry once, a stable sort of the row by ITS column block
    hub = util.decode_long_cb(plan)
    assert len(hub) == plan.stats["lcb_rows"]
    for r, ent in hub.items():
        k = np.argsort(ci[rp[r]:rp[r + 1]] // plan.stats["lcb_col_block"], kind="stable")
        assert [c for c, _ in ent] == ci[rp[r]:rp[r + 1]][k].tolist() and np.array_equal(np.asarray([x for _, x in ent], np.float16), v[rp[r]:rp[r + 1]][k]), r
    assert sorted(rows) == sorted(int(np.flatnonzero(order == r)[0]) if not natural else r for r in range(m) if lens[r] > 0 and r not in hub)
    for pos, (cs, vs) in rows.items():
        r = pos if natural else order[pos]
        # a row's entries: the column blocks in ascending order, CSR order inside a block -- i.e. a stable sort of the row by column block
        cb = plan.stats["tp_col_block"]
        k = np.argsort(ci[rp[r]:rp[r + 1]] // cb, kind="stable")
        assert cs == ci[rp[r]:rp[r + 1]][k].tolist(), (pos, r)
        assert np.array_equal(np.asarray(vs, np.float16), v[rp[r]:rp[r + 1]][k]), (pos, r)


@pytest.mark.parametrize("natural", [False, True])
@pytest.mark.parametrize("m,n,seed,cb,rb", [(3000, 2500, 5, 0, 0), (3000, 2500, 6, 512, 64), (700, 70000, 7, 32768, 100), (64, 8, 8, 8, 1), (1, 5, 9, 0, 0)])
def test_streams_decode_to_the_csr(dasp, oracle, m, n, seed, cb, rb, natural):
    rp, ci, v = util.mixed_matrix(m, n, seed, values="f16", dtype=np.float16)
    plan = dasp.Plan(rp, ci, v, n, precision=16, two_phase=1, tp_col_block=cb, tp_row_block=rb, y_order=dasp.Y_NATURAL if natural else dasp.Y_PERMUTED)
    st = plan.stats
    assert st["two_phase"] == 1 and st["n_col_panels"] == 0 and st["tp_col_block"] == (cb or 32768)
    assert st["fill0_nnz_reg"] == st["tp_segments"] * st["tp_seg_elems"] and st["fill0_nnz_reg"] + st["lcb_elems"] >= st["nnzA"]
    assert abs(st["rate_fill0"] - (st["tp_segments"] * st["tp_seg_elems"] + st["lcb_elems"] - st["nnzA"]) / max(st["nnzA"], 1)) < 1e-12
    # the whole-matrix classifier: the reference's counters and permutation do not depend on the form
    P = oracle.Packed(16, rp, ci, v.astype(np.float64), n)
    for f in "row_long row_block row_zero short_row_1 short_row_2 short_row_3 short_row_4 common_13".split():
        assert st[f] == getattr(P, f), f
    assert (plan.order_rid == P.order_rid).all()
    row0 = plan.host_array("tp_rb_row0")
    assert row0[0] == 0 and row0[-1] == m and (np.diff(row0) >= 1).all() and (np.diff(row0) <= (rb or 4096)).all()
    _check_decodes(plan, rp, ci, v, m, natural)
    plan.close()


@pytest.mark.parametrize("natural", [False, True])
def test_hybrid_hub_rows_leave_the_streams(dasp, natural, tmp_path):
    """r6: the rows of >= max(block_longest, 64 x column blocks) nonzeros of a two-phase plan go column-blocked (Plan::lcb, the kernels of the column
    panels' hub rows) when they hold >= a quarter of the nonzeros; long_cb = -1 keeps every row in the streams; fewer than that: no hybrid; the plan file round-trips"""
    lens = [5] * 700 + [300, 2999, 0, 256, 255, 1200] + [17] * 500 + [1] * 300          # 4755 of 17310 nonzeros in the four rows of >= 256
    rp, ci, v = util.csr_from_lengths(lens, 3000, 17, values="f16", dtype=np.float16)
    m = len(lens)
    kw = dict(precision=16, two_phase=1, y_order=dasp.Y_NATURAL if natural else dasp.Y_PERMUTED)
    plan = dasp.Plan(rp, ci, v, 3000, **kw)
    st = plan.stats
    assert st["two_phase"] == 1 and st["lcb_rows"] == 4 and st["lcb_col_block"] == 32768 and st["lcb_elems"] % 128 == 0
    assert set(util.decode_long_cb(plan)) == {i for i, L in enumerate(lens) if L >= 256}
    _check_decodes(plan, rp, ci, v, m, natural)
    dst = plan.host_array("lcb_row_dst")
    rid = plan.host_array("lcb_row_id")
    assert (dst == rid).all() if natural else (plan.order_rid[dst] == rid).all()
    path = str(tmp_path / "hyb.plan")
    plan.save(path)
    again = dasp.Plan.load(path)
    assert again.stats["lcb_rows"] == 4 and util.decode_long_cb(again) == util.decode_long_cb(plan) and util.decode_plan(again) == util.decode_plan(plan)
    again.close()
    # an entry lost from the hub rows' stream: the loader adds the nonzeros up
    raw = bytearray(open(path, "rb").read())
    lcol = plan.host_array("lcb_lcol")
    at = bytes(raw).find(lcol.tobytes())
    k = int(np.flatnonzero(lcol != 0xFFFF)[0])
    assert at > 0
    raw[at + 2 * k: at + 2 * k + 2] = (0xFFFF).to_bytes(2, "little")
    (tmp_path / "bad.plan").write_bytes(bytes(raw))
    with pytest.raises(dasp.DaspError) as e:
        dasp.Plan.load(str(tmp_path / "bad.plan"))
    assert "add up" in str(e.value)
    plan.close()
    off = dasp.Plan(rp, ci, v, 3000, long_cb=-1, **kw)
    assert off.stats["lcb_rows"] == 0 and off.stats["tp_segments"] * 64 >= ci.size
    _check_decodes(off, rp, ci, v, m, natural)
    off.close()
    few = [5] * 20000 + [300]                                       # one hub row with 0.3 % of the nonzeros: not worth two more launches
    rp2, ci2, v2 = util.csr_from_lengths(few, 3000, 18, values="f16", dtype=np.float16)
    assert dasp.Plan(rp2, ci2, v2, 3000, **kw).stats["lcb_rows"] == 0
    assert dasp.Plan(rp2, ci2, v2, 3000, long_cb=1, **kw).stats["lcb_rows"] == 1          # forced


def test_row_blocks_balance_the_nonzeros_of_the_sorted_order(dasp):
    """the permuted order is sorted by row length: blocks of equal row counts would give the first workgroups a hundred times the work of the last"""
    rng = np.random.default_rng(3)
    lens = np.minimum(rng.zipf(1.6, 60000), 3000)
    rp, ci, v = util.csr_from_lengths(lens, 50000, 4, values="f16", dtype=np.float16)
    plan = dasp.Plan(rp, ci, v, 50000, precision=16, two_phase=1)
    row0, seg0 = plan.host_array("tp_rb_row0"), plan.host_array("tp_rb_seg0")
    per_block = np.diff(seg0) * plan.stats["tp_seg_elems"]
    target = max(16384, min(1 << 17, ci.size // 1024 + 1))
    assert per_block.max() <= 2 * target + 3000 + 64 * plan.host_array("tp_unit").reshape(-1, 3)[:, 0].max()      # one row past the target + the tiles' padding
    assert (np.diff(row0) <= 4096).all()
    plan.close()


def test_forced_on_what_it_cannot_do_is_an_error(dasp):
    rp, ci, v = util.mixed_matrix(200, 300, 1)
    with pytest.raises(dasp.DaspError):
        dasp.Plan(rp, ci, v, 300, precision=64, two_phase=1)                     # f16 only
    v16 = v.astype(np.float16)
    with pytest.raises(dasp.DaspError):
        dasp.Plan(rp, ci, v16, 300, precision=16, two_phase=1, tp_col_block=12)   # not a multiple of 8
    with pytest.raises(dasp.DaspError):
        dasp.Plan(rp, ci, v16, 300, precision=16, two_phase=1, tp_row_block=9000)
    # auto never picks it for a small matrix, off is off
    assert dasp.Plan(rp, ci, v16, 300, precision=16).stats["two_phase"] == 0
    assert dasp.Plan(rp, ci, v16, 300, precision=16, two_phase=-1).stats["two_phase"] == 0


def test_empty_and_degenerate_inputs(dasp):
    for m, n in ((0, 0), (0, 7), (5, 0), (5, 7)):
        rp = np.zeros(m + 1, np.int32)
        plan = dasp.Plan(rp, np.zeros(0, np.int32), np.zeros(0, np.float16), n, precision=16, two_phase=1)
        st = plan.stats
        assert st["two_phase"] == 1 and st["tp_segments"] == 0 and st["tp_units"] == 0 and st["tp_row_blocks"] == (1 if m else 0)
        assert util.decode_plan(plan) == {}
        plan.close()


def test_plan_file_round_trip_and_validator(dasp, tmp_path):
    rp, ci, v = util.mixed_matrix(2500, 2000, 11, values="f16", dtype=np.float16)
    plan = dasp.Plan(rp, ci, v, 2000, precision=16, two_phase=1, tp_col_block=256, tp_row_block=128, long_cb=-1)
    path = str(tmp_path / "tp.plan")
    plan.save(path)
    again = dasp.Plan.load(path)
    assert again.stats["two_phase"] == 1 and again.stats["tp_segments"] == plan.stats["tp_segments"]
    for name in ("tp_rb_row0", "tp_rb_seg0", "tp_unit", "tp_dst", "tp_lcol", "tp_lrow", "tp_val"):
        assert np.array_equal(again.host_array(name), plan.host_array(name)), name
    _check_decodes(again, rp, ci, v, 2500, False)
    again.close()
    # a corrupted stream must be refused, not uploaded: a local row beyond its row block, a local column beyond its column block, dst not a permutation
    raw = bytearray(open(path, "rb").read())
    S = plan.stats["tp_segments"]
    lrow = plan.host_array("tp_lrow")
    k = int(np.flatnonzero(lrow != 0xFFFF)[0])
    blob = lrow.tobytes()
    at = bytes(raw).find(blob)
    assert at > 0
    bad = bytearray(raw)
    bad[at + 2 * k: at + 2 * k + 2] = (5000).to_bytes(2, "little")
    (tmp_path / "bad1.plan").write_bytes(bytes(bad))
    with pytest.raises(dasp.DaspError) as e:
        dasp.Plan.load(str(tmp_path / "bad1.plan"))
    assert "local row" in str(e.value)
    dst = plan.host_array("tp_dst")
    at = bytes(raw).find(dst.tobytes())
    assert at > 0 and S > 2
    bad = bytearray(raw)
    bad[at: at + 4] = int(dst[1]).to_bytes(4, "little")
    (tmp_path / "bad2.plan").write_bytes(bytes(bad))
    with pytest.raises(dasp.DaspError) as e:
        dasp.Plan.load(str(tmp_path / "bad2.plan"))
    assert "permutation" in str(e.value)
    plan.close()


def test_automatic_forms_by_size(dasp):
    """the automatic rules' size gates (r5, tools/size_sweep.py): the two-phase form from ~10 M nonzeros on for an f16 matrix whose rows scatter; two column panels for the
    sake of the column-blocked hub rows when those hold most of a >= 16 M-nonzero f64 matrix, even though x fits an XCD's L2"""
    rng = np.random.default_rng(5)
    def graph(m, per_row, hubs=0, hub_len=0):
        lens = np.full(m, per_row, np.int64)
        if hubs:
            lens[rng.choice(m, hubs, replace=False)] = hub_len
        rp = np.zeros(m + 1, np.int64); np.cumsum(lens, out=rp[1:])
        ci = rng.integers(0, m, int(rp[-1]), dtype=np.int32)
        return rp.astype(np.int32), ci
    rp, ci = graph(700000, 16)                                       # 11.2 M nonzeros, uniform columns
    assert dasp.Plan(rp, ci, np.ones(ci.size, np.float16), 700000, precision=16).stats["two_phase"] == 1
    rp, ci = graph(700000, 13)                                       # 9.1 M: the plain kernels
    assert dasp.Plan(rp, ci, np.ones(ci.size, np.float16), 700000, precision=16).stats["two_phase"] == 0
    rp, ci = graph(300000, 24, hubs=2000, hub_len=5000)              # 17.2 M, 58 % of them in 2000 hub rows; x = 2.4 MB
    st = dasp.Plan(rp, ci, np.ones(ci.size), 300000, precision=64).stats
    assert st["n_col_panels"] == 2 and st["lcb_rows"] == 2000, st
    rp, ci = graph(300000, 50, hubs=600, hub_len=5000)               # 18 M, 17 % in hub rows: nothing to block, nothing to stage
    st = dasp.Plan(rp, ci, np.ones(ci.size), 300000, precision=64).stats
    assert st["n_col_panels"] == 0 and st["lcb_rows"] == 0, st


def test_column_block_halves_when_the_columns_are_skewed(dasp):
    """late r5: automatic tp_col_block -- 32768, or 16384 when one block of 32768 columns holds more than four times its share of the nonzeros (R-MAT graphs: rmat_2M f16
    0.0893 -> 0.0826 ms; the even families prefer the larger block); an explicit tp_col_block is taken as given"""
    rng = np.random.default_rng(9)
    m, n, per = 40000, 300000, 8
    rp = (np.arange(m + 1) * per).astype(np.int32)
    even = rng.integers(0, n, m * per).astype(np.int32)
    skew = np.where(rng.random(m * per) < 0.5, rng.integers(0, 32768, m * per), rng.integers(0, n, m * per)).astype(np.int32)     # half of the nonzeros in the first block of ten
    v = np.ones(m * per, np.float16)
    assert dasp.Plan(rp, even, v, n, precision=16, two_phase=1).stats["tp_col_block"] == 32768
    sk = dasp.Plan(rp, skew, v, n, precision=16, two_phase=1)
    assert sk.stats["tp_col_block"] == 16384
    got = util.decode_plan(sk)
    assert sum(len(c) for c, _ in got.values()) == skew.size
    assert dasp.Plan(rp, skew, v, n, precision=16, two_phase=1, tp_col_block=32768).stats["tp_col_block"] == 32768

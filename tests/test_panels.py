"""Column panels (dasp_options_t::col_panels): host-side structure.  The parent keeps the whole matrix's order_rid and the
reference's classifier counters; the panels are natural-order plans over disjoint column ranges whose union is A."""
import numpy as np
import pytest

import util

REF_COUNTERS = ("short_row_1", "common_13", "short_row_3", "short_row_4", "short_row_2", "row_long", "row_block", "row_zero",
                "nnz_short", "nnz_long", "rowloop", "data_origin1")


def rows_of(plan, m):
    """row -> sorted [(col, val)] decoded from the packed arrays of a single (non-panel) plan"""
    dec = util.decode_plan(plan)
    order = plan.order_rid
    out = [[] for _ in range(m)]
    for slot, (cs, vs) in dec.items():
        out[int(order[slot])] = sorted(zip(cs, [float(x) for x in vs]))
    return out


@pytest.mark.parametrize("prec", [64, 16])
@pytest.mark.parametrize("y_order", [0, 1])
@pytest.mark.parametrize("P,tile,lcb", [(2, 0, 0), (3, -1, -1), (7, 5, 1), (3, 32, 0)])
def test_panels_partition_the_matrix(dasp, prec, y_order, P, tile, lcb):
    dt = np.float64 if prec == 64 else np.float16
    m, n = 1500, 5000
    lens = np.random.default_rng(3).choice([0, 1, 2, 3, 4, 9, 40, 300, 700], size=m, p=[.05, .15, .1, .15, .1, .2, .15, .07, .03])
    rp, ci, v = util.csr_from_lengths(lens, n, 5, dtype=dt)
    single = dasp.Plan(rp, ci, v, n, precision=prec, y_order=y_order, col_panels=1)
    plan = dasp.Plan(rp, ci, v, n, precision=prec, y_order=y_order, col_panels=P, row_tile_max=tile, long_cb=lcb)      # 0 = auto: 16 (f16) / 8 (f64)
    assert single.n_panels == 0 and single.stats["n_col_panels"] == 0
    assert plan.n_panels == P and plan.stats["n_col_panels"] == P
    np.testing.assert_array_equal(plan.order_rid, single.order_rid)
    for k in REF_COUNTERS:
        assert plan.stats[k] == single.stats[k], k
    assert plan.host_array("med_val").size == 0 and plan.host_array("piece_ptr").size == 0
    want = [sorted(zip(ci[rp[r]:rp[r + 1]].tolist(), v[rp[r]:rp[r + 1]].astype(np.float64).tolist())) for r in range(m)]
    got = [[] for _ in range(m)]
    prev_end = 0
    for k in range(P):
        sub, cb, ce = plan.panel(k)
        assert cb == prev_end and cb % 64 == 0 and ce > cb
        prev_end = ce
        assert sub.stats["n_col_panels"] == 0
        dmap = sub.host_array("dst_map")
        if y_order == 0:
            np.testing.assert_array_equal(dmap[plan.order_rid], np.arange(m))     # row -> its slot in the parent
        else:
            assert dmap.size == 0
        nnz_k = 0
        for r, ent in enumerate(rows_of(sub, m)):
            assert all(cb <= c < ce for c, _ in ent)
            got[r] += ent
            nnz_k += len(ent)
        assert nnz_k == sub.stats["nnzA"] + sub.stats["row_tile_nnz"]      # (the panels keep their short rows in row tiles)
        assert (sub.stats["row_tile_nnz"] > 0) == (tile >= 0) and sub.stats["row_tile_max"] == (0 if tile < 0 else tile or (16 if prec == 16 else 8)) and sub.stats["n_row_tiles"] == (-(-m // 64) if tile >= 0 else 0)
    assert prev_end == n
    # column-blocked long rows (long_cb; auto: the rows of >= 256 hold more than a quarter of this matrix): they are in no panel, every entry once in the piece streams
    hub = util.decode_long_cb(plan)
    assert (len(hub) > 0) == (lcb >= 0) and set(hub) == ({r for r in range(m) if lens[r] >= 256} if lcb >= 0 else set())
    n_hub = 0
    for r, ent in hub.items():
        assert got[r] == []
        # a row's entries: column blocks in ascending order, CSR order inside a block
        k = np.argsort(ci[rp[r]:rp[r + 1]] // plan.stats["lcb_col_block"], kind="stable")
        assert [c for c, _ in ent] == ci[rp[r]:rp[r + 1]][k].tolist()
        got[r] = ent
        n_hub += len(ent)
    assert [sorted(g) for g in got] == want
    assert sum(plan.panel(k)[0].stats["nnzA"] + plan.panel(k)[0].stats["row_tile_nnz"] for k in range(P)) + n_hub == ci.size
    assert plan.stats["row_tile_nnz"] == sum(plan.panel(k)[0].stats["row_tile_nnz"] for k in range(P))


def test_empty_panels_are_dropped_and_auto_stays_off_for_small_inputs(dasp):
    rp, ci, v = util.mixed_matrix(800, 1000, 4)
    ci = (ci % 400).astype(np.int32)                       # nothing in the upper columns
    for r in range(800):                                   # keep rows sorted (duplicates are fine)
        ci[rp[r]:rp[r + 1]].sort()
    plan = dasp.Plan(rp, ci, v, 1000, col_panels=2)
    assert plan.n_panels == 1 and plan.panel(0)[1:] == (0, 512)
    with pytest.raises(dasp.DaspError):
        plan.panel(1)
    assert dasp.Plan(rp, ci, v, 1000).n_panels == 0        # auto: far below the size where cache blocking pays
    with pytest.raises(dasp.DaspError):
        dasp.Plan(rp, ci, v, 1000, col_panels=65)
    empty = dasp.Plan(np.zeros(6, np.int32), np.zeros(0, np.int32), np.zeros(0), 10, col_panels=4)
    assert empty.n_panels == 0


def test_panels_follow_the_partitioned_x_layout(dasp):
    m, n = 600, 1000
    rp, ci, v = util.mixed_matrix(m, n, 8)
    bounds, stride = np.array([0, 300, 1000], np.int32), 768
    plan = dasp.Plan(rp, ci, v, n, col_panels=3, part_bounds=bounds, part_stride=stride, y_order=1)
    assert plan.x_len == 2 * stride and plan.n_panels >= 2
    remap = np.where(ci < 300, ci, stride + ci - 300)
    seen = 0
    for k in range(plan.n_panels):
        sub, cb, ce = plan.panel(k)
        assert sub.stats["colA"] == 2 * stride
        for r, ent in enumerate(rows_of(sub, m)):
            cols = {c for c, _ in ent}
            assert cols <= set(remap[rp[r]:rp[r + 1]].tolist()) and all(cb <= c < ce for c in cols)
            seen += len(ent)
    assert seen == ci.size


@pytest.mark.parametrize("prec", [64, 16])
def test_panel_plan_file_round_trip(dasp, prec, tmp_path):
    dt = np.float64 if prec == 64 else np.float16
    rp, ci, v = util.mixed_matrix(900, 3000, 12, dtype=dt)
    plan = dasp.Plan(rp, ci, v, 3000, precision=prec, col_panels=3)
    path = tmp_path / "panels.plan"
    plan.save(path)
    back = dasp.Plan.load(path)
    assert back.n_panels == 3 and back.stats == plan.stats
    np.testing.assert_array_equal(back.order_rid, plan.order_rid)
    for k in range(3):
        a, b = plan.panel(k), back.panel(k)
        assert a[1:] == b[1:] and a[0].stats == b[0].stats
        for name in ("dst_map", "order", "med_ptr", "med_val", "med_cid", "irr_val", "irr_cid", "long_val", "long_cid", "piece_dst",
                     "short_val", "short_cid"):
            np.testing.assert_array_equal(a[0].host_array(name), b[0].host_array(name), err_msg=name)


def test_long_cb_leaves_something_for_the_panels(dasp, tmp_path):
    """a matrix whose every row is a hub (found by the sanitizer driver, r5): with all rows column-blocked no panel would hold a nonzero and the parent would have no
    panels at all -- the rule then keeps the rows in the panels; a plan with SOME hub rows saves, loads and decodes"""
    lens = [300] * 40
    rp, ci, v = util.csr_from_lengths(lens, 4000, 3)
    plan = dasp.Plan(rp, ci, v, 4000, col_panels=2, long_cb=1)
    assert plan.stats["lcb_rows"] == 0 and plan.n_panels == 2
    lens = [300] * 40 + [7] * 100
    rp, ci, v = util.csr_from_lengths(lens, 4000, 3)
    plan = dasp.Plan(rp, ci, v, 4000, col_panels=2, long_cb=1)
    assert plan.stats["lcb_rows"] == 40 and plan.n_panels == 2 and plan.stats["lcb_elems"] % 128 == 0
    path = str(tmp_path / "lcb.plan")
    plan.save(path)
    again = dasp.Plan.load(path)
    assert again.stats["lcb_rows"] == 40 and util.decode_long_cb(again) == util.decode_long_cb(plan)
    for name in ("lcb_ptr", "lcb_unit", "lcb_row_dst", "lcb_row_id", "lcb_lcol", "lcb_val"):
        assert np.array_equal(again.host_array(name), plan.host_array(name)), name

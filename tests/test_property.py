"""Property-based checks of the host half of the plan (no GPU): for arbitrary small matrices and option corners, the
native packed format decodes to exactly the CSR rows, and categories / output permutation equal the oracle's."""
import os

import numpy as np
import pytest
from hypothesis import HealthCheck, given, settings, strategies as st

import util

LENGTH_POOL = [0, 0, 1, 1, 2, 3, 3, 4, 5, 6, 8, 15, 16, 17, 31, 32, 33, 63, 64, 65, 127, 255, 256, 257, 300, 1023, 1024, 1025, 2100]


@st.composite
def matrices(draw):
    m = draw(st.integers(0, 260))
    n = draw(st.sampled_from([1, 2, 7, 64, 1000, 70000, 200000]))
    seed = draw(st.integers(0, 2 ** 31 - 1))
    rng = np.random.default_rng(seed)
    heavy = draw(st.sampled_from(["mixed", "short", "medium", "pairs"]))
    pool = {"mixed": LENGTH_POOL, "short": [0, 1, 2, 3, 4, 1, 3], "medium": [5, 6, 7, 9, 12, 20, 40, 80, 200, 255],
            "pairs": [1, 3] * 6 + [2, 4, 9]}[heavy]
    lens = rng.choice(pool, size=m) if m else np.zeros(0, np.int64)
    if heavy == "pairs" and m:
        lens = np.concatenate([lens, rng.choice([1, 3], size=300)])       # enough len-1 / len-3 rows for common_13 > 0
    rp = np.zeros(lens.size + 1, np.int64)
    np.cumsum(lens, out=rp[1:])
    band = draw(st.sampled_from([0, 50, 5000]))
    if band and lens.size:
        rows = np.repeat(np.arange(lens.size), lens)
        ci = np.clip(rows * max(1, n // max(lens.size, 1)) + rng.integers(-band, band + 1, rows.size), 0, n - 1)
    else:
        ci = rng.integers(0, n, int(rp[-1]))
    v = rng.integers(1, 9, int(rp[-1])).astype(np.float64) / 4.0           # exactly representable in f16
    return rp.astype(np.int32), ci.astype(np.int32), v, n


OPTS = st.fixed_dictionaries(dict(
    threshold=st.sampled_from([0.75, 0.75, 0.3, 1.0]), block_longest=st.sampled_from([256, 256, 64, 16, 700]),
    long_piece=st.sampled_from([0, 64, 256, 4096]), x_window=st.sampled_from([0, -1, 2048, 100000]), x_window_hybrid=st.sampled_from([0, 1, -1]),
    row_window=st.sampled_from([0, 64, 256, 1024]), cid16=st.sampled_from([0, -1, 1]), y_order=st.sampled_from([0, 1]),
    slab_max_len=st.sampled_from([0, 0, 4, 9, 32]), piece_min_len=st.sampled_from([0, 0, -1, 7, 60]), chunk_pairs=st.sampled_from([0, 0, -1, 1, 2]), cid8=st.sampled_from([0, 0, -1])))


@settings(max_examples=150, deadline=None, suppress_health_check=[HealthCheck.function_scoped_fixture, HealthCheck.too_slow])
@given(mat=matrices(), opts=OPTS, prec=st.sampled_from([64, 16]))
def test_packed_format_round_trips(dasp, oracle, mat, opts, prec):
    rp, ci, v, n = mat
    m = rp.size - 1
    dt = np.float64 if prec == 64 else np.float16
    plan = dasp.Plan(rp, ci, v.astype(dt), n, precision=prec, **opts)
    # categories and the output permutation: the oracle's (reference geometry), whatever the native options are
    P = oracle.Packed(prec, rp, ci, v, n, threshold=opts["threshold"], block_longest=opts["block_longest"])
    st_ = plan.stats
    for f in "row_long row_block row_zero short_row_1 short_row_2 short_row_3 short_row_4 common_13 nnz_short nnz_long".split():
        assert st_[f] == getattr(P, f), f
    order = plan.order_rid
    assert (order == P.order_rid).all()
    # the packed arrays hold every nonzero of every row exactly once, in row order
    rows = util.decode_plan(plan)
    assert sorted(rows) == list(range(m))
    for slot in range(m):
        r = order[slot]
        cs, vs = rows[slot]
        assert cs == ci[rp[r]:rp[r + 1]].tolist()
        assert np.array_equal(np.asarray(vs, np.float64), v[rp[r]:rp[r + 1]])
    stored = st_["fill0_nnz_short"] + st_["fill0_nnz_long"] + st_["fill0_nnz_reg"] + st_["nnz_irreg"]
    assert stored >= st_["nnzA"]


@settings(max_examples=60, deadline=None, suppress_health_check=[HealthCheck.function_scoped_fixture, HealthCheck.too_slow])
@given(mat=matrices(), opts=OPTS, prec=st.sampled_from([64, 16]), panels=st.integers(2, 6))
def test_column_panels_partition_arbitrary_matrices(dasp, mat, opts, prec, panels):
    """col_panels: the parent keeps the single plan's order_rid; the panels' packed arrays hold every nonzero exactly once,
    each inside its panel's column range."""
    rp, ci, v, n = mat
    m = rp.size - 1
    dt = np.float64 if prec == 64 else np.float16
    single = dasp.Plan(rp, ci, v.astype(dt), n, precision=prec, col_panels=1, **opts)
    plan = dasp.Plan(rp, ci, v.astype(dt), n, precision=prec, col_panels=panels, **opts)
    assert (plan.order_rid == single.order_rid).all()
    if ci.size == 0 or m == 0:
        assert plan.n_panels == 0
        return
    got = [[] for _ in range(m)]
    for k in range(plan.n_panels):
        sub, cb, ce = plan.panel(k)
        order = sub.order_rid
        for slot, (cs, vs) in util.decode_plan(sub).items():
            assert all(cb <= c < ce for c in cs)
            got[int(order[slot])] += list(zip(cs, [float(x) for x in vs]))
    for r, ent in util.decode_long_cb(plan).items():          # hub rows of a panel plan live in the column-blocked piece streams (long_cb auto), in no panel
        assert got[r] == []
        got[r] = ent
    for r in range(m):
        assert sorted(got[r]) == sorted(zip(ci[rp[r]:rp[r + 1]].tolist(), v[rp[r]:rp[r + 1]].astype(dt).astype(np.float64).tolist()))


@settings(max_examples=80, deadline=None, suppress_health_check=[HealthCheck.function_scoped_fixture, HealthCheck.too_slow])
@given(mat=matrices(), y_order=st.sampled_from([0, 1]), cb=st.sampled_from([0, 8, 64, 4096, 65536]), rb=st.sampled_from([0, 1, 7, 300, 8192]), longest=st.sampled_from([256, 16]))
def test_two_phase_streams_decode_for_arbitrary_matrices(dasp, mat, y_order, cb, rb, longest):
    """the two-phase form of an f16 plan (r5): whatever the matrix and the block sizes, the tile streams hold every nonzero once, at its row's output position, with
    the whole matrix's order_rid -- and a plan file of it loads (the validator accepts what the packer wrote)"""
    import tempfile
    rp, ci, v, n = mat
    m = rp.size - 1
    plan = dasp.Plan(rp, ci, v.astype(np.float16), n, precision=16, two_phase=1, tp_col_block=cb, tp_row_block=rb, y_order=y_order, block_longest=longest)
    single = dasp.Plan(rp, ci, v.astype(np.float16), n, precision=16, two_phase=-1, col_panels=1, y_order=y_order, block_longest=longest)
    assert plan.stats["two_phase"] == 1 and (plan.order_rid == single.order_rid).all()
    order = plan.order_rid
    got = util.decode_plan(plan)
    hub = util.decode_long_cb(plan)          # r6, the hybrid: hub rows (when they hold >= a quarter of the nonzeros) live column-blocked beside the streams, in no tile
    for pos, (cs, vs) in got.items():
        r = pos if y_order == 1 else order[pos]
        assert r not in hub
        assert sorted(zip(cs, vs)) == sorted(zip(ci[rp[r]:rp[r + 1]].tolist(), v[rp[r]:rp[r + 1]].astype(np.float16).astype(np.float64).tolist()))
    for r, ent in hub.items():
        assert sorted(ent) == sorted(zip(ci[rp[r]:rp[r + 1]].tolist(), v[rp[r]:rp[r + 1]].astype(np.float16).astype(np.float64).tolist()))
    assert sum(len(c) for c, _ in got.values()) + sum(len(e) for e in hub.values()) == ci.size
    with tempfile.TemporaryDirectory() as d:
        plan.save(os.path.join(d, "p.plan"))
        assert dasp.Plan.load(os.path.join(d, "p.plan")).stats["tp_segments"] == plan.stats["tp_segments"]


@pytest.mark.gpu
@settings(max_examples=int(os.environ.get("DASP_HYP_EXAMPLES", "80")), deadline=None, suppress_health_check=[HealthCheck.function_scoped_fixture, HealthCheck.too_slow])
@given(mat=matrices(), opts=OPTS, prec=st.sampled_from([64, 16]), from_device=st.booleans(), panels=st.sampled_from([1, 1, 2, 3, 5]), form=st.sampled_from([0, 0, 1, 2]))
def test_spmv_matches_csr_for_arbitrary_matrices(dasp, oracle, mat, opts, prec, from_device, panels, form):
    import torch
    rp, ci, v, n = mat
    m = rp.size - 1
    dt = np.float64 if prec == 64 else np.float16
    tdt = torch.float64 if prec == 64 else torch.float16
    xh = (np.random.default_rng(int(rp[-1]) + m).integers(1, 9, n) / 8.0).astype(dt)
    opts = dict(opts)
    if form == 1 and prec == 16:
        opts.update(two_phase=1, tp_col_block=[0, 8, 512][m % 3], tp_row_block=[0, 3, 100][n % 3])       # r5: the two-phase form, odd block sizes included
    elif form == 2:
        opts.update(long_cb=1)                                                                               # r5: hub rows of a column-panel plan column-blocked
    if from_device and m > 0:
        d = [torch.from_numpy(a).cuda() for a in (rp, ci if ci.size else np.zeros(1, np.int32), v.astype(dt) if v.size else np.zeros(1, dt))]
        plan = dasp.Plan.from_device(d[0].data_ptr(), d[1].data_ptr(), d[2].data_ptr(), m, n, int(rp[-1]), precision=prec, **opts)
    else:
        plan = dasp.Plan(rp, ci, v.astype(dt), n, precision=prec, col_panels=panels, **opts).upload()
    x = torch.from_numpy(xh).cuda()
    y = torch.full((max(m, 1),), float("nan"), dtype=tdt, device="cuda")
    plan.spmv(x.data_ptr(), y.data_ptr(), torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    got = y[:m].double().cpu().numpy()
    ref = oracle.csr_spmv(rp, ci, v, xh.astype(np.float64))
    scale = np.maximum(oracle.csr_absrow(rp, ci, v, xh.astype(np.float64)), 1e-300)
    perm = plan.order_rid if opts["y_order"] == 0 else np.arange(m)
    assert np.isfinite(got).all()
    assert (np.abs(got - ref[perm]) / scale[perm]).max(initial=0.0) <= (1e-12 if prec == 64 else 1e-2)
    plan.close()

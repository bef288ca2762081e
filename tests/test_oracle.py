"""The oracle against its pins: the reference's own mmio.h (oracle/_ref), the radix-sort
known-answer vector recorded from the real code, the committed fixtures,
and the reference's accounting / all-ones identities."""
import ctypes as C
import json
import os

import numpy as np
import pytest

import util

GOLD = os.path.join(os.path.dirname(__file__), "golden")
EXP = json.load(open(os.path.join(GOLD, "expected.json")))
MTX = sorted(k for k in EXP if k.endswith(".mtx"))


def test_radix_known_answer(oracle):
    ka = EXP["radix_known_answer"]
    k, i = oracle.radix_sort_desc(ka["key"], ka["idx"])
    assert k.tolist() == ka["key_sorted"] and i.tolist() == ka["idx_sorted"]


def test_radix_is_stable_descending(oracle):
    rng = np.random.default_rng(3)
    key = rng.integers(5, 256, 5000)
    k, i = oracle.radix_sort_desc(key, np.arange(key.size))
    ref = np.argsort(-key, kind="stable")
    assert (i == ref).all() and (k == key[ref]).all()


def test_exclusive_scan(oracle):
    assert oracle.exclusive_scan([3, 1, 4, 1, 5]).tolist() == [0, 3, 4, 8, 9]
    assert oracle.exclusive_scan([7]).tolist() == [7]      # len 1 is left untouched (mmio_highlevel.h:12-13)
    assert oracle.exclusive_scan([]).tolist() == []


@pytest.mark.parametrize("name", MTX)
def test_banner_and_size_match_reference_mmio(oracle, name):
    """oracle restatement == the reference's own mmio.h compiled into oracle/_ref"""
    R = oracle.ref_mmio()
    if R is None:
        pytest.skip("oracle/_ref not built (reference tree absent and no prebuilt copy)")
    p = os.path.join(GOLD, name).encode()
    tc = C.create_string_buffer(4)
    rc = R.ref_mm_read_banner_path(p, tc)
    assert (rc, tc.raw.decode("latin1")) == oracle.mm_read_banner(p.decode())
    M, N, nz = C.c_int(), C.c_int(), C.c_int()
    rc2 = R.ref_mm_read_size_path(p, C.byref(M), C.byref(N), C.byref(nz))
    assert (rc2, M.value, N.value, nz.value) == oracle.mm_read_size(p.decode())


def _coo_matches_reference(oracle, path):
    """file-order COO triples: the REFERENCE's mm_read_mtx_crd_data and mm_read_mtx_crd_entry (src/mmio.h:866-980, compiled
    into oracle/_ref from where the file lies) == the oracle's restatement of mmio_allinone's entry loop
    (src/mmio_highlevel.h:663-697).  Returns the number of entries compared, or None when the reference rejects the file."""
    ref = oracle.ref_read_crd(path)
    if ref is None:
        pytest.skip("oracle/_ref not built (reference tree absent and no prebuilt copy)")
    rc, tc, M, N, I, J, re, im = oracle.mm_read_coo(path)
    if rc != 0:
        return None
    brc, bI, bJ, bval = ref["bulk"]
    erc, eI, eJ, ere, eim = ref["entries"]
    if tc[2] == "I":
        # the reference's mmio.h parsers do not take integer matrices (MM_UNSUPPORTED_TYPE = 15): mmio_allinone reads them with its
        # own fscanf("%d %d %d") (mmio_highlevel.h:680-684), which only the restatement covers
        assert brc == 15 and erc == 15
        return 0
    assert brc == 0 and erc == 0 and ref["nz"] == I.size
    assert (bI - 1 == I).all() and (bJ - 1 == J).all() and (eI - 1 == I).all() and (eJ - 1 == J).all()
    if tc[2] == "C":
        assert (bval[0::2] == re).all() and (bval[1::2] == im).all() and (ere == re).all() and (eim == im).all()
    elif tc[2] == "R":
        assert (bval[: I.size] == re).all() and (ere == re).all()
    else:
        assert (re == 1.0).all()                              # pattern: the parsers leave val alone, mmio_allinone stores 1.0 (:692)
    return int(I.size)


@pytest.mark.parametrize("name", MTX)
def test_coo_entries_match_reference_mmio(oracle, name):
    if EXP[name]["rc"] != 0:
        pytest.skip("file the loader rejects")
    if name == "multi_per_line.mtx":
        # fscanf-token semantics on a file whose entries straddle lines: both sides read whitespace-separated tokens
        pass
    n = _coo_matches_reference(oracle, os.path.join(GOLD, name))
    assert n is not None


@pytest.mark.parametrize("field", ["real", "pattern", "complex"])
@pytest.mark.parametrize("symm", ["general", "symmetric", "hermitian", "skew-symmetric"])
def test_coo_entries_match_reference_mmio_random_files(oracle, tmp_path, field, symm):
    from test_loader import write_mtx
    rng = np.random.default_rng(hash((field, symm, 1)) % 2 ** 32)
    m = n = 300
    k = 4000
    rows = rng.integers(0, m, k)
    cols = rng.integers(0, n, k)
    if symm != "general":
        lo = np.minimum(rows, cols)
        rows, cols = np.maximum(rows, cols), lo
    p = str(tmp_path / "r.mtx")
    write_mtx(p, m, n, rows, cols, rng.uniform(-50, 50, k), field, symm, rng)
    assert _coo_matches_reference(oracle, p) == k


@pytest.mark.parametrize("name", MTX)
def test_loader_fixtures(oracle, name):
    e = EXP[name]
    rc, m, n, nnz, sym, rp, ci, v = oracle.mmio_allinone(os.path.join(GOLD, name))
    assert rc == e["rc"]
    assert oracle.mm_read_banner(os.path.join(GOLD, name)) == (e["banner_rc"], e["typecode"])
    if rc == 0:
        assert (m, n, nnz, sym) == (e["m"], e["n"], e["nnz"], e["sym"])
        assert rp.tolist() == e["row_ptr"] and ci.tolist() == e["col_idx"] and v.tolist() == e["val"]


def test_symmetric_4x4_matches_survey_run(oracle):
    """CSR the surveyor obtained from the real loader on a 4x4 symmetric file"""
    s = EXP["sym4_survey"]
    rc, m, n, nnz, sym, rp, ci, v = oracle.mmio_allinone(os.path.join(GOLD, s["file"]))
    assert rc == 0 and rp.tolist() == s["row_ptr"] and ci.tolist() == s["col_idx"]


@pytest.mark.parametrize("prec", [64, 16])
@pytest.mark.parametrize("builder,m,n,seed", [(util.mixed_matrix, 3000, 2500, 7), (util.pair_heavy_matrix, 4000, 3000, 11)])
def test_pack_identities_and_eval(oracle, prec, builder, m, n, seed):
    rp, ci, v = builder(m, n, seed)
    P = oracle.Packed(prec, rp, ci, v, n)
    lens = np.diff(rp)
    # the reference's accounting identity (dasp_f64.h:1091)
    assert P.nnz_short + P.nnz_long + P.origin_nnz_reg + P.nnz_irreg == P.nnzA
    # order_rid is a permutation, categories in the documented order
    assert sorted(P.order_rid.tolist()) == list(range(m))
    assert (lens[P.order_rid[: P.row_long]] >= 256).all()
    ml = lens[P.order_rid[P.row_long: P.row_long + P.row_block]]
    assert ((ml >= 5) & (ml < 256)).all() and (np.diff(ml) <= 0).all()
    assert (lens[P.order_rid[m - P.row_zero:]] == 0).all()
    # evaluation of the packed format == CSR product
    x = np.random.default_rng(5).uniform(-1, 1, n)
    y = P.eval(x)
    yref = oracle.csr_spmv(rp, ci, v, x)
    scale = oracle.csr_absrow(rp, ci, v, x)
    assert (np.abs(y - yref[P.order_rid]) <= 1e-12 * np.maximum(scale[P.order_rid], 1e-300)).all()
    # all-ones identity of the reference's driver: y[i] == nnz(row order_rid[i]) exactly
    P1 = oracle.Packed(prec, rp, ci, np.ones_like(v), n)
    assert (P1.eval(np.ones(n)) == lens[P1.order_rid]).all()


def test_packer_regression_values(oracle):
    for tag, e in EXP["packer_cases"].items():
        builder = {"mixed": util.mixed_matrix, "pairs": util.pair_heavy_matrix}[e["builder"]]
        rp, ci, v = builder(e["m"], e["n"], e["seed"])
        P = oracle.Packed(int(tag[-2:]), rp, ci, v, e["n"])
        for f, want in e.items():
            if f in ("builder", "m", "n", "seed"):
                continue
            got = "%016x" % oracle.fnv1a_i32(P.order_rid) if f == "order_fnv" else int(getattr(P, f))
            assert got == want, (tag, f)


def test_round_f16(oracle):
    a = np.array([0.0, 1.0, 1.0009765625, 1.00048828125, 65504.0, 1e-8, 3.14159, -2.71828])
    assert (oracle.round_f16(a) == a.astype(np.float16).astype(np.float64)).all()


HAND = json.load(open(os.path.join(GOLD, "handworked.json")))


@pytest.mark.parametrize("case", ["small24", "pairs300"])
@pytest.mark.parametrize("prec", [64, 16])
def test_handworked_classifier_and_order(oracle, dasp, case, prec):
    """Counters and order_rid worked out by hand from the reference's source lines (tests/golden/make_handworked.py holds the
    derivation) == the oracle's restatement == the product's classifier (plan.cpp): a pin that trusts neither."""
    h = HAND[case]
    lens = np.asarray(h["lengths"])
    n = 400
    rp, ci, v = util.csr_from_lengths(lens, n, 5)
    want_c = h.get("counters") or h["counters_f%d" % prec]
    want_o = h["order_f%d" % prec]
    P = oracle.Packed(prec, rp, ci, v, n, block_longest=h["block_longest"])
    assert P.order_rid.tolist() == want_o
    for k, val in want_c.items():
        assert int(getattr(P, k)) == val, (k, "oracle")
    for k, val in h.get("packer_f%d" % prec, {}).items():              # padded sizes and pointer arrays of the reference-geometry packers
        got = getattr(P, k)
        assert (got.tolist() if isinstance(val, list) else int(got)) == val, (k, "oracle packer")
    plan = dasp.Plan(rp, ci, v.astype(np.float64 if prec == 64 else np.float16), n, precision=prec, block_longest=h["block_longest"])
    assert plan.order_rid.tolist() == want_o
    st = plan.stats
    for k, val in want_c.items():
        assert st[k] == val, (k, "product")
    # the product also reports the reference's padded sizes (dasp_stats_t::ref_*): the same hand-derived numbers
    for k, val in h.get("packer_f%d" % prec, {}).items():
        if not isinstance(val, list):
            assert st["ref_" + k] == val, (k, "product, reference geometry")
    plan.close()


@pytest.mark.parametrize("seed,m,n", [(1, 3000, 2500), (2, 500, 4000), (3, 1, 7), (4, 4000, 4000)])
def test_csr_product_against_scipy(oracle, seed, m, n):
    """The y oracle is the builder's own serial loop (the reference has no CPU SpMV): pin it to an INDEPENDENT implementation --
    scipy.sparse's CSR product -- on seeded matrices with every row category, duplicates and unsorted columns included.  The loop
    adds a row's products in storage order (as the reference's tail loop does, dasp_f64.h:189-192) and scipy's kernel adds them in
    the same order, so the results agree exactly wherever no duplicate was summed beforehand; we demand 1e-15 of sum |a x|."""
    import scipy.sparse as sp
    rp, ci, v = util.mixed_matrix(m, n, seed)
    x = np.random.default_rng(seed + 100).uniform(-1, 1, n)
    A = sp.csr_matrix((v, ci, rp), shape=(m, n))                          # keeps duplicates and the storage order
    want = A @ x
    absA = sp.csr_matrix((np.abs(v), ci, rp), shape=(m, n))
    scale = absA @ np.abs(x)
    got = oracle.csr_spmv(rp, ci, v, x)
    got_abs = oracle.csr_absrow(rp, ci, v, x)
    assert np.all(np.abs(got - want) <= 1e-15 * np.maximum(scale, 1e-300))
    assert np.all(np.abs(got_abs - scale) <= 1e-15 * np.maximum(scale, 1e-300))
    assert np.array_equal(got == 0, want == 0) or np.all(scale[(got == 0) != (want == 0)] > 0)     # empty rows are exact zeros in both
    e = np.diff(rp) == 0
    assert np.all(got[e] == 0) and np.all(got_abs[e] == 0)


def _fnv_words(a):
    """the survey's hash of order_rid: FNV-1a constants over the int WORDS (h ^= (unsigned)order_rid[i]; h *= prime)"""
    h = 1469598103934665603
    for w in np.asarray(a).astype(np.uint32).tolist():
        h = ((h ^ w) * 1099511628211) & 0xFFFFFFFFFFFFFFFF
    return "%016x" % h


@pytest.mark.parametrize("prec", [64, 16])
def test_survey_recorded_reference_outputs(oracle, dasp, tmp_path, prec):
    """The one place where numbers produced BY THE REFERENCE'S OWN HOST CODE exist for the hot path: SURVEY.md section 8(c) / App. D.3 record what
    the reference's classifier and packers (dasp_f64.h:486-1157, dasp_f16.h:1015-1449) printed for a 3000 x 3000 pattern matrix -- the
    classifier tuple, the padded sizes and (f64) a hash of order_rid.  The matrix is committed as data (survey_g3000.mtx.gz), the numbers are
    copied from the survey.  The oracle's restatement, the product's loader + classifier and the product's reference-geometry sizes
    (dasp_stats_t::ref_*) must all reproduce them: a pin of rows a4-a11 to an execution of the reference, recorded by the survey."""
    import gzip
    want_all = json.load(open(os.path.join(GOLD, "survey_g3000.json")))
    path = tmp_path / "survey_g3000.mtx"
    path.write_bytes(gzip.open(os.path.join(GOLD, "survey_g3000.mtx.gz")).read())
    rc, m, n, nnz, sym, rp, ci, v = oracle.mmio_allinone(str(path))
    assert (rc, m, n, nnz, sym) == (0, want_all["rows"], want_all["cols"], want_all["nnz"], 0)
    want = dict(want_all["f%d" % prec])
    fnv = want.pop("order_rid_fnv", None)
    rate = want.pop("rate_fill0_3dp", None)
    P = oracle.Packed(prec, rp, ci, np.ones(nnz), n)
    for k, val in want.items():
        assert int(getattr(P, k)) == val, (k, "oracle")
    if fnv:
        assert _fnv_words(P.order_rid) == fnv
    if rate is not None:
        assert round(P.rate_fill0, 3) == rate
    # the product: its own loader on the same file, its classifier, its order_rid, and the reference-geometry sizes it reports
    m2, n2, nnz2, sym2, rp2, ci2, v2 = dasp.mmio_allinone(str(path), prec)
    assert (m2, n2, nnz2) == (m, n, nnz) and np.array_equal(rp2, rp) and np.array_equal(ci2, ci)
    plan = dasp.Plan(rp2, ci2, v2, n2, precision=prec)
    st = plan.stats
    for k, val in want.items():
        key = k if k in st and not k.startswith("fill0") and k not in ("nnz_irreg", "blocknum", "warp_number") else "ref_" + k
        assert st[key] == val, (k, "product")
    if fnv:
        assert _fnv_words(plan.order_rid) == fnv
    if rate is not None:
        assert round(st["ref_rate_fill0"], 3) == rate
    assert np.array_equal(plan.order_rid, P.order_rid)
    plan.close()

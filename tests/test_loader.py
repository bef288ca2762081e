"""Product loader (dasp_amd/csrc/mmio.cpp through the C ABI) == oracle restatement of
mmio_allinone, on the committed fixtures and on generated files (all field types, symmetry,
duplicates, empty rows); f16 values are the correctly rounded binary16 of the parsed double."""
import json
import os

import numpy as np
import pytest

GOLD = os.path.join(os.path.dirname(__file__), "golden")
EXP = json.load(open(os.path.join(GOLD, "expected.json")))
MTX = sorted(k for k in EXP if k.endswith(".mtx"))


@pytest.mark.parametrize("name", MTX)
def test_fixture(dasp, name):
    e = EXP[name]
    path = os.path.join(GOLD, name)
    if e["rc"] != 0:
        with pytest.raises(dasp.DaspError) as err:
            dasp.mmio_allinone(path)
        assert err.value.status == e["rc"]       # -2 banner / -4 size, as mmio_highlevel.h:626-640
        return
    m, n, nnz, sym, rp, ci, v = dasp.mmio_allinone(path, 64)
    assert (m, n, nnz, sym) == (e["m"], e["n"], e["nnz"], e["sym"])
    assert rp.tolist() == e["row_ptr"] and ci.tolist() == e["col_idx"] and v.tolist() == e["val"]
    m, n, nnz, sym, rp, ci, h = dasp.mmio_allinone(path, 16)
    assert h.dtype == np.float16 and (h == np.asarray(e["val"]).astype(np.float16)).all()


def write_mtx(path, m, n, rows, cols, vals, field, symm, rng):
    with open(path, "w") as f:
        f.write("%%%%MatrixMarket matrix coordinate %s %s\n%% generated\n%d %d %d\n" % (field, symm, m, n, len(rows)))
        for i, j, v in zip(rows, cols, vals):
            if field == "pattern":
                f.write("%d %d\n" % (i + 1, j + 1))
            elif field == "integer":
                f.write("%d %d %d\n" % (i + 1, j + 1, int(v)))
            elif field == "complex":
                f.write("%d %d %.17g %.17g\n" % (i + 1, j + 1, v, rng.uniform()))
            else:
                f.write("%d %d %s\n" % (i + 1, j + 1, rng.choice(["%.17g", "%.6e", "%+.3f", "%g"]) % v))


@pytest.mark.parametrize("field", ["real", "pattern", "integer", "complex"])
@pytest.mark.parametrize("symm", ["general", "symmetric", "hermitian", "skew-symmetric"])
def test_random_files_match_oracle(dasp, oracle, tmp_path, field, symm):
    rng = np.random.default_rng(hash((field, symm)) % 2 ** 32)
    m = n = 300
    k = 4000
    rows = rng.integers(0, m, k)
    cols = rng.integers(0, n, k)
    if symm != "general":
        lo = np.minimum(rows, cols)
        rows, cols = np.maximum(rows, cols), lo          # lower triangle, diagonal and duplicates included
    vals = rng.uniform(-50, 50, k)
    p = str(tmp_path / "r.mtx")
    write_mtx(p, m, n, rows, cols, vals, field, symm, rng)
    rc, om, on, onnz, osym, orp, oci, ov = oracle.mmio_allinone(p)
    assert rc == 0
    gm, gn, gnnz, gsym, grp, gci, gv = dasp.mmio_allinone(p, 64)
    assert (gm, gn, gnnz, gsym) == (om, on, onnz, osym)
    assert (grp == orp).all() and (gci == oci).all()
    assert (gv == ov).all()                               # bit-exact doubles
    *_, hv = dasp.mmio_allinone(p, 16)
    assert (hv == ov.astype(np.float16)).all()


def test_entry_errors(dasp, tmp_path):
    p = tmp_path / "short.mtx"
    p.write_text("%%MatrixMarket matrix coordinate real general\n3 3 3\n1 1 1.0\n2 2 2.0\n")
    with pytest.raises(dasp.DaspError) as e:
        dasp.mmio_allinone(str(p))
    assert e.value.status == -5
    p.write_text("%%MatrixMarket matrix coordinate real general\n3 3 1\n4 1 1.0\n")
    with pytest.raises(dasp.DaspError) as e:
        dasp.mmio_allinone(str(p))
    assert e.value.status == -5


def test_size_line_larger_than_the_file_allocates_nothing(dasp, tmp_path):
    """a tiny file whose size line promises 2 G entries is an entry error, decided before any nz-sized allocation"""
    import resource
    p = tmp_path / "liar.mtx"
    p.write_text("%%MatrixMarket matrix coordinate real general\n1 1 2000000000\n1 1 1.0\n")
    before = resource.getrusage(resource.RUSAGE_SELF).ru_maxrss
    with pytest.raises(dasp.DaspError) as e:
        dasp.mmio_allinone(str(p))
    assert e.value.status == -5
    assert resource.getrusage(resource.RUSAGE_SELF).ru_maxrss - before < 200 * 1024      # KiB: nowhere near the 32 GB the line asks for


@pytest.mark.parametrize("prec", [64, 16])
def test_binary_csr_cache_round_trip(dasp, tmp_path, prec):
    m, n, nnz, sym, rp, ci, v = dasp.mmio_allinone(os.path.join(GOLD, "sym_real.mtx"), prec)
    p = str(tmp_path / "a.csrbin")
    dasp.csr_save(p, rp, ci, v, n, sym, prec)
    m2, n2, nnz2, sym2, rp2, ci2, v2 = dasp.csr_load(p, prec)
    assert (m2, n2, nnz2, sym2) == (m, n, nnz, sym)
    assert (rp2 == rp).all() and (ci2 == ci).all() and v2.dtype == v.dtype and (v2 == v).all()
    with pytest.raises(dasp.DaspError):
        dasp.csr_load(p, 16 if prec == 64 else 64)          # wrong precision is refused
    with open(p, "r+b") as f:
        f.truncate(40)
    with pytest.raises(dasp.DaspError):
        dasp.csr_load(p, prec)

"""Consumes tests/golden/ref_f64.json / ref_f16.json -- what tools/ref_dump.cu.txt prints when it runs the CUDA REFERENCE on this repository's
fixtures -- and compares the product with it: CSR hashes, every counter and padded size of the reference's CSV row
(dasp_f64.h:1439-1441; the padded sizes against dasp_stats_t::ref_*), order_rid, and y (GPU).  Until somebody with a CUDA box commits such a
file these tests skip, and parity of the packers and of y stays "partial" (DESIGN.md section 2)."""
import json
import os
import re

import numpy as np
import pytest

import util

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def _load(prec):
    path = os.path.join(GOLD, "ref_f%d.json" % prec)
    if not os.path.exists(path):
        pytest.skip("no %s: run tools/ref_dump.cu.txt on a CUDA box (parity of packers / y stays 'partial' until then)" % os.path.basename(path))
    txt = open(path).read()
    txt = re.sub(r'"log": "[^"]*"', '"log": ""', txt.replace("\n", " "))          # spmv_all's own stdout lines land inside "log"
    return json.loads(txt)


def _mtx(name, tmp):
    """a fixture by the base name the dump recorded: tests/golden/<name>, or <name>.gz unpacked on the fly (survey_g3000.mtx travels gzipped)"""
    path = os.path.join(GOLD, name)
    if os.path.exists(path):
        return path
    if os.path.exists(path + ".gz"):
        import gzip
        out = os.path.join(str(tmp), name)
        with gzip.open(path + ".gz", "rb") as f, open(out, "wb") as g:
            g.write(f.read())
        return out
    pytest.fail("the reference dump names %s, which tests/golden does not hold" % name)


def _fnv(a):
    h = 1469598103934665603
    for b in np.ascontiguousarray(a, np.int32).view(np.uint8).tolist():
        h = ((h ^ b) * 1099511628211) & 0xFFFFFFFFFFFFFFFF
    return "%016x" % h


CSV_INT = "rowA colA nnzA short_row_1 common_13 short_row_3 short_row_4 short_row_2 row_long row_block nnz_short ref_fill0_nnz_short nnz_long ref_fill0_nnz_long ref_origin_nnz_reg ref_fill0_nnz_reg ref_nnz_irreg".split()


@pytest.mark.parametrize("prec", [64, 16])
def test_counters_sizes_and_order_match_the_cuda_reference(dasp, prec, tmp_path):
    ref = _load(prec)
    assert ref["precision"] == prec
    for mat in ref["matrices"]:
        path = _mtx(mat["file"], tmp_path)
        m, n, nnz, sym, rp, ci, v = dasp.mmio_allinone(path, precision=prec)
        assert (m, n, nnz) == (mat["rowA"], mat["colA"], mat["nnzA"])
        assert _fnv(rp) == mat["rowptr_fnv"] and _fnv(ci) == mat["colidx_fnv"], mat["file"]           # the loader's CSR, bit for bit
        for run in mat["runs"]:
            vals = np.ones(nnz, v.dtype) if run["mode"] == "ones" else v
            plan = dasp.Plan(rp, ci, vals, n, precision=prec)
            st = plan.stats
            cols = run["csv_row"].split(",")
            for k, name in enumerate(CSV_INT):
                assert int(cols[1 + k]) == st[name], (mat["file"], name)
            assert abs(float(cols[18]) - st["ref_rate_fill0"]) < 1e-6 and int(cols[20]) == st["ref_data_X"], mat["file"]
            assert plan.order_rid.tolist() == run["order_rid"], mat["file"]
            plan.close()


@pytest.mark.gpu
@pytest.mark.parametrize("prec", [64, 16])
def test_y_matches_the_cuda_reference(dasp, torch_cuda, prec, tmp_path):
    """Y_val of the reference's spmv_all, slot by slot: exact in the all-ones mode, 1e-12 (f64) / 1e-2 (f16: the reference accumulates in
    half, this build in f32) of sum |a x| otherwise.  f16: the Y_val the reference leaves behind is its bypass kernel's (dasp_spmv2, the run its "SpMV_X2" line times:
    dasp_f16.h:1636-1705), in both value modes."""
    torch = torch_cuda
    ref = _load(prec)
    dt, tdt = (np.float64, torch.float64) if prec == 64 else (np.float16, torch.float16)
    for mat in ref["matrices"]:
        m, n, nnz, sym, rp, ci, v = dasp.mmio_allinone(_mtx(mat["file"], tmp_path), precision=prec)
        for run in mat["runs"]:
            ones = run["mode"] == "ones"
            vals = np.ones(nnz, dt) if ones else v
            xh = np.ones(n, dt) if ones else (1.0 + (np.arange(n) % 7) / 8.0).astype(dt)
            plan = dasp.Plan(rp, ci, vals, n, precision=prec).upload()
            x = torch.from_numpy(xh).cuda()
            y = torch.zeros(max(m, 1), dtype=tdt, device="cuda")
            plan.spmv(x.data_ptr(), y.data_ptr(), 0)
            torch.cuda.synchronize()
            got = y[:m].double().cpu().numpy()
            want = np.asarray(run["y_permuted"], np.float64)
            rows = plan.order_rid
            scale = np.array([np.abs(vals[rp[r]:rp[r + 1]].astype(np.float64) * xh[ci[rp[r]:rp[r + 1]]].astype(np.float64)).sum() for r in rows])
            tol = (0.0 if ones and prec == 64 else 1e-12 if prec == 64 else 1e-2)
            assert np.all(np.abs(got - want) <= tol * np.maximum(scale, 1e-300) + (0 if prec == 64 else 1e-3)), mat["file"]
            plan.close()

"""GPU parity tests (run with -m gpu on an MI355X): the HIP path, called through the C ABI,
against the CPU oracle on the same seeded inputs.
  f64: |y - y_csr| <= 1e-12 * sum_j |a_ij x_j|   (BASELINE.json north_star: 1e-12 relative)
  f16: |y - y_csr| <= 1e-2  * sum_j |a_ij x_j|   (north_star: 1e-2; f16 inputs, f32 accumulate, f16 store)
  A == 1, x == 1 (the reference driver's mode): y[i] == nnz(row order_rid[i]) EXACTLY.
"""
import os
import subprocess
import sys

import numpy as np
import pytest

import util

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
TOL = {64: 1e-12, 16: 1e-2}


def tdtype(torch, prec):
    return torch.float64 if prec == 64 else torch.float16


def run_spmv(torch, plan, xh, rows, prec):
    x = torch.from_numpy(np.ascontiguousarray(xh)).cuda()
    y = torch.full((max(rows, 1),), float("nan"), dtype=tdtype(torch, prec), device="cuda")   # every slot must be written
    plan.spmv(x.data_ptr(), y.data_ptr(), torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    return y[:rows].double().cpu().numpy()


def check(oracle, dasp, torch, rp, ci, v, n, prec, **kw):
    dt = np.float64 if prec == 64 else np.float16
    v = v.astype(dt)
    rng = np.random.default_rng(99)
    xh = (rng.uniform(-1, 1, n) if prec == 64 else rng.uniform(0.5, 1.5, n)).astype(dt)
    m = rp.size - 1
    ref = oracle.csr_spmv(rp, ci, v.astype(np.float64), xh.astype(np.float64))
    scale = np.maximum(oracle.csr_absrow(rp, ci, v.astype(np.float64), xh.astype(np.float64)), 1e-300)
    for y_order in (dasp.Y_PERMUTED, dasp.Y_NATURAL):
        plan = dasp.Plan(rp, ci, v, n, precision=prec, y_order=y_order, **kw).upload()
        got = run_spmv(torch, plan, xh, m, prec)
        perm = plan.order_rid if y_order == dasp.Y_PERMUTED else np.arange(m)
        err = np.abs(got - ref[perm]) / scale[perm]
        assert np.isfinite(got).all()
        assert err.max() <= TOL[prec], (prec, y_order, err.max(), int(err.argmax()))
        plan.close()
    # the reference driver's mode: all-ones values and x -> exact row lengths
    lens = np.diff(rp)
    plan = dasp.Plan(rp, ci, np.ones(ci.size, dt), n, precision=prec, **kw).upload()
    got = run_spmv(torch, plan, np.ones(n, dt), m, prec)
    want = lens[plan.order_rid].astype(np.float64)
    if prec == 16:
        want = want.astype(np.float16).astype(np.float64)       # rows longer than 2048 round on the f16 store
    assert (got == want).all()
    plan.close()


def test_mfma_lane_maps(dasp, torch_cuda):
    dasp.selftest_mfma()


@pytest.mark.parametrize("prec", [64, 16])
@pytest.mark.parametrize("tag,builder,m,n,seed", [
    ("mixed", util.mixed_matrix, 3000, 2500, 7),
    ("pairs", util.pair_heavy_matrix, 4000, 3000, 11),
    ("tiny", util.mixed_matrix, 37, 50, 3),
    ("one_row", util.mixed_matrix, 1, 10, 5),
])
def test_parity_seeded(oracle, dasp, torch_cuda, prec, tag, builder, m, n, seed):
    rp, ci, v = builder(m, n, seed, values="f16" if prec == 16 else "uniform")
    check(oracle, dasp, torch_cuda, rp, ci, v, n, prec)


@pytest.mark.parametrize("prec", [64, 16])
def test_long_rows_cut_into_pieces(oracle, dasp, torch_cuda, prec):
    """rows longer than long_piece exercise the partial sums + dasp_long_reduce_kernel"""
    lens = [5000, 256, 1023, 1024, 1025, 4096, 300, 7, 2, 20000, 0, 700]
    rp, ci, v = util.csr_from_lengths(lens, 30000, 4, values="f16" if prec == 16 else "uniform")
    check(oracle, dasp, torch_cuda, rp, ci, v, 30000, prec)
    check(oracle, dasp, torch_cuda, rp, ci, v, 30000, prec, long_piece=256)


@pytest.mark.parametrize("prec", [64, 16])
@pytest.mark.parametrize("lens", [[5] * 100, [255] * 33, [17, 16, 16, 5], [4] * 1000, [1] * 300 + [3] * 300, [0] * 70,
                                   [2] * 129, [3] * 257, [6, 0, 6, 0, 1]])
def test_single_category_edges(oracle, dasp, torch_cuda, prec, lens):
    rp, ci, v = util.csr_from_lengths(lens, 997, 13, values="f16" if prec == 16 else "uniform")
    check(oracle, dasp, torch_cuda, rp, ci, v, 997, prec)


@pytest.mark.parametrize("prec", [64, 16])
def test_empty_matrix(dasp, torch_cuda, prec):
    rp = np.zeros(9, np.int32)
    plan = dasp.Plan(rp, np.zeros(0, np.int32), np.zeros(0), 4, precision=prec).upload()
    got = run_spmv(torch_cuda, plan, np.ones(4, np.float64 if prec == 64 else np.float16), 8, prec)
    assert (got == 0).all()


@pytest.mark.parametrize("name,prec,scale", [
    ("cop20k_A", 64, 0.1), ("nlpkkt160", 64, 0.004), ("powerlaw_1M", 64, 0.05), ("HV15R", 64, 0.01),
    ("Queen_4147", 64, 0.005), ("webbase-1M", 16, 0.1), ("ljournal-2008", 16, 0.01), ("rmat_2M", 64, 0.02), ("rmat_2M", 16, 0.02),
])
def test_parity_synthetic_standins(oracle, dasp, torch_cuda, name, prec, scale):
    rows, cols = dasp.synth_dims(name, scale)
    rp, ci = dasp.synth_csr(name, scale)
    rng = np.random.default_rng(1)
    v = rng.uniform(0.5, 1.5, ci.size) if prec == 16 else rng.uniform(-1, 1, ci.size)
    check(oracle, dasp, torch_cuda, rp, ci, v, cols, prec)


@pytest.mark.parametrize("name,prec", [("HV15R", 64), ("ljournal-2008", 16), ("Queen_4147", 64), ("nlpkkt160", 64), ("webbase-1M", 16),
                                       ("HV15R-unstructured", 64), ("cop20k_A", 64), ("powerlaw_1M", 64), ("rmat_2M", 64), ("rmat_2M", 16),
                                       ("ljournal-2008-uniform", 16), ("webbase-1M-uniform", 16)])
def test_full_size_random_x_parity(oracle, dasp, torch_cuda, name, prec):
    """BASELINE's full sizes (scale 1.0): seeded random values and x, >= 100 k sampled rows (the 4096 longest + a uniform sample)
    against the oracle at the north_star tolerance -- bench.py's verified_random_x, the check the all-ones mode cannot make"""
    import importlib.util
    spec = importlib.util.spec_from_file_location("bench", os.path.join(ROOT, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    rows, cols = dasp.synth_dims(name, 1.0)
    rp, ci = dasp.synth_csr(name, 1.0)
    res = bench.verify_random_x(torch_cuda, dasp, oracle, rp, ci, cols, prec)
    assert res["ok"] and res["rows_checked"] >= 100000 and res["max_rel_err"] <= TOL[prec], res


@pytest.mark.parametrize("name,scale", [("cop20k_A", 1.0), ("HV15R", 0.02), ("powerlaw_1M", 0.05), ("webbase-1M", 0.2), ("nlpkkt160", 0.01)])
def test_vendor_comparator_agrees_with_dasp(torch_cuda, name, scale):
    """the reference's comparator check (src/main_f64.cu:3-16 verify_new: y_cusparse[order_rid[i]] against y_dasp[i]) with rocSPARSE's CSR SpMV
    in cuSPARSE's place: seeded random values and x, every row, 1e-12 relative to sum |a_ij x_j| (the reference: 1e-5 absolute)"""
    import re
    exe = os.path.join(ROOT, "dasp_amd", "bin", "dasp_rocsparse")
    if not os.path.exists(exe):
        pytest.skip("dasp_rocsparse was not built (no librocsparse at build time)")
    r = subprocess.run([exe, name, repr(scale), "3", "1", "1"], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    assert "mismatches=0 (vs dasp_plan_spmv through order_rid" in r.stdout
    m = re.search(r"compare: rows=(\d+) max_rel_err=([0-9.e+-]+)", r.stdout)
    assert m and int(m.group(1)) > 1000 and float(m.group(2)) <= 1e-12


def test_padded_slots_do_not_read_x0(oracle, dasp, torch_cuda):
    """the reference's padded slots multiply 0 by x[0] (dasp_f64.h:1127-1128): x[0] = inf poisons
    unrelated rows there.  Here pads never touch x."""
    rp, ci, v = util.mixed_matrix(2000, 1500, 17)
    ci = np.where(ci == 0, 1, ci).astype(np.int32)            # no row references column 0
    x = np.random.default_rng(1).uniform(-1, 1, 1500)
    x[0] = np.inf
    plan = dasp.Plan(rp, ci, v, 1500).upload()
    got = run_spmv(torch_cuda, plan, x, 2000, 64)
    assert np.isfinite(got).all()


def test_one_shot_spmv_all(oracle, dasp, torch_cuda, capfd):
    """spmv_all with the reference's argument list: Y in permuted order + order_rid, result line on stdout"""
    for prec in (64, 16):
        dt = np.float64 if prec == 64 else np.float16
        rp, ci, v = util.mixed_matrix(1500, 1200, 23, values="ones")
        y, order = dasp.spmv_all("mixed.mtx", v.astype(dt), rp, ci, np.ones(1200, dt), 1500, 1200, int(rp[-1]), 4, 0.75, 256, precision=prec)
        P = oracle.Packed(prec, rp, ci, v, 1200)
        assert (order == P.order_rid).all()
        want = np.diff(rp)[order].astype(dt).astype(np.float64)
        assert (y.astype(np.float64) == want).all()
    out = capfd.readouterr().out
    assert out.count("SpMV_X:") == 2 and out.count("SpMV_X2:") == 1     # dasp_f64.h:1398 ; dasp_f16.h:1717-1718 (f16 prints both)


def test_partitioned_x_layout(oracle, dasp, torch_cuda):
    """row slice + column remap: x is read from an all-gather-shaped buffer of padded slices"""
    torch = torch_cuda
    rp, ci, v = util.mixed_matrix(900, 900, 31)
    bounds = dasp.partition_rows(rp, 3)
    stride = int(np.diff(bounds).max()) + 5
    x = np.random.default_rng(2).uniform(-1, 1, 900)
    xg = np.zeros(3 * stride)
    for g in range(3):
        xg[g * stride: g * stride + bounds[g + 1] - bounds[g]] = x[bounds[g]:bounds[g + 1]]
    ref = oracle.csr_spmv(rp, ci, v, x)
    scale = np.maximum(oracle.csr_absrow(rp, ci, v, x), 1e-300)
    for g in range(3):
        r0, r1 = bounds[g], bounds[g + 1]
        sl = slice(rp[r0], rp[r1])
        plan = dasp.Plan(rp[r0:r1 + 1] - rp[r0], ci[sl], v[sl], 900, y_order=dasp.Y_NATURAL, part_bounds=bounds, part_stride=stride).upload()
        got = run_spmv(torch, plan, xg, r1 - r0, 64)
        assert (np.abs(got - ref[r0:r1]) / scale[r0:r1]).max() <= 1e-12


def test_timing_protocol(dasp, torch_cuda):
    torch = torch_cuda
    rp, ci, v = util.mixed_matrix(20000, 20000, 41)
    plan = dasp.Plan(rp, ci, v, 20000).upload()
    x = torch.ones(20000, dtype=torch.float64, device="cuda")
    y = torch.zeros(20000, dtype=torch.float64, device="cuda")
    wall, ev = plan.time(x.data_ptr(), y.data_ptr(), torch.cuda.current_stream().cuda_stream, warmup=10, iters=50)
    assert 0 < ev < 50 and 0 < wall < 50


def test_spmv_before_upload_is_an_error(dasp, torch_cuda):
    rp, ci, v = util.mixed_matrix(10, 10, 1)
    plan = dasp.Plan(rp, ci, v, 10)
    with pytest.raises(dasp.DaspError) as e:
        plan.spmv(1, 1)
    assert e.value.status == -22


@pytest.mark.parametrize("exe,fixture", [("dasp_f64", "sym_real.mtx"), ("dasp_f16", "gen_real.mtx")])
def test_cli_drivers(torch_cuda, oracle, exe, fixture, tmp_path):
    """spmv_double / spmv_half equivalents: load .mtx, all-ones, spmv_all, verify through order_rid; the CSV record is the
    reference's: spmv_all's partial row completed by the driver with the comparator columns and a newline (main_f64.cu:151-153,
    main_f16.cu:148-150), one row per run"""
    (tmp_path / "data").mkdir()
    for run in range(2):
        r = subprocess.run([os.path.join(ROOT, "dasp_amd", "bin", exe), os.path.join(ROOT, "tests", "golden", fixture)],
                           cwd=tmp_path, capture_output=True, text=True, timeout=300)
        assert r.returncode == 0, r.stdout + r.stderr
        assert "compute succeed" in r.stdout and "SpMV_X:" in r.stdout
    csv = (tmp_path / "data" / ("spmv_f64_record.csv" if exe == "dasp_f64" else "spmv_f16_record.csv")).read_text()
    assert csv.endswith("\n")
    rows = csv.splitlines()
    assert len(rows) == 2
    ncol = {"dasp_f64": 18 + 7 + 5, "dasp_f16": 18 + 10 + 6}[exe]       # spmv_all's columns + main()'s tail
    for row in rows:
        assert row.startswith(os.path.join(ROOT, "tests", "golden", fixture) + ",")
        assert len(row.split(",")) == ncol and not row.endswith(",")
    # the padded-size columns hold the REFERENCE's geometry (what the CUDA reference writes for this file: dasp_f64.h:1439-1441), i.e. the
    # oracle's reference-geometry packer on the same CSR with the driver's all-ones values; the native sizes have a file of their own
    prec = 64 if exe == "dasp_f64" else 16
    rc, m, n, nnz, sym, rp, ci, v = oracle.mmio_allinone(os.path.join(ROOT, "tests", "golden", fixture))
    assert rc == 0
    P = oracle.Packed(prec, rp, ci, np.ones(ci.size), n)
    c = rows[0].split(",")
    assert [int(c[k]) for k in (1, 2, 3)] == [m, n, nnz]
    assert [int(c[k]) for k in range(4, 18)] == [P.short_row_1, P.common_13, P.short_row_3, P.short_row_4, P.short_row_2, P.row_long, P.row_block,
                                                  P.nnz_short, P.fill0_nnz_short, P.nnz_long, P.fill0_nnz_long, P.origin_nnz_reg, P.fill0_nnz_reg, P.nnz_irreg]
    assert abs(float(c[18]) - P.rate_fill0) < 1e-6 and int(c[19]) == 256 and int(c[20]) == P.data_X
    native = (tmp_path / "data" / ("dasp_amd_native_f%d.csv" % prec)).read_text().splitlines()
    assert len(native) == 2 and len(native[0].split(",")) == 10


@pytest.mark.parametrize("args", [("HV15R", "0.02", "64"), ("HV15R", "0.02", "16"), ("cop20k_A", "0.5", "64"), ("webbase-1M", "0.2", "16"),
                                  ("nlpkkt160", "0.01", "64", "5", "2", "0.75", "1024", "0", "0", "1", "0", "0", "0", "0", "0", "2")])
def test_dasp_bench_checks_values_and_columns(torch_cuda, args):
    """the A/B driver of tools/ab.sh verifies what it times: the all-ones product exactly (row lengths) and a product with
    x[j] = 1 + (j % 61) / 64 against a host CSR loop -- an all-ones product alone cannot see a wrong column id.  Last case: 16-bit
    ids with every pairing mode forced (chunk_pairs = 2)"""
    a = list(args) + ["5", "2"] if len(args) == 3 else list(args)
    r = subprocess.run([os.path.join(ROOT, "dasp_amd", "bin", "dasp_bench")] + a, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "mismatches=0" in r.stdout.splitlines()[-1], r.stdout


def test_bench_reads_real_matrices_from_dasp_mtx_dir(dasp, torch_cuda, tmp_path):
    """DASP_MTX_DIR: bench.py takes <workload>.mtx from that directory (through the product loader + CSR cache) instead of the
    seeded stand-in -- here a MatrixMarket file written from the small stand-in itself, so the result is known"""
    import json
    import sys
    rows, cols = dasp.synth_dims("HV15R", 0.01)
    rp, ci = dasp.synth_csr("HV15R", 0.01)
    with open(tmp_path / "HV15R.mtx", "w") as f:
        f.write("%%%%MatrixMarket matrix coordinate pattern general\n%d %d %d\n" % (rows, cols, ci.size))
        r = np.repeat(np.arange(rows), np.diff(rp))
        np.savetxt(f, np.stack([r + 1, ci + 1], 1), fmt="%d")
    env = dict(os.environ, DASP_MTX_DIR=str(tmp_path))
    out = None
    for run in range(2):                                   # second run: from the binary CSR cache written by the first
        p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "5", "--warmup", "2", "--no-suite", "--no-vendor",
                            "--no-cpu-baseline"], env=env, capture_output=True, text=True, timeout=600)
        assert p.returncode == 0, p.stdout[-2000:] + p.stderr[-2000:]
        out = json.loads([l for l in p.stdout.splitlines() if l.startswith("{")][-1])
        assert out["data"] == "suitesparse" and out["verified"] is True and out["verified_random_x"]["ok"] is True
        assert out["config"]["rows"] == rows and out["config"]["nnz"] == ci.size and "DASP_MTX_DIR" in out["config"]["workload"]
    assert (tmp_path / "HV15R.mtx.f64.csrbin").exists()


@pytest.mark.parametrize("prec", [64, 16])
@pytest.mark.parametrize("kw", [dict(), dict(x_window=21000, row_window=64), dict(x_window=100000, row_window=128), dict(row_window=512), dict(row_window=1024), dict(x_window=-1)])
def test_lds_staged_x_windows(oracle, dasp, torch_cuda, prec, kw):
    """narrow-band matrix: windows of rows share a span of x that is staged in LDS (auto on); small caps mix LDS and
    global-gather windows in one launch; results and the output permutation are the same in every mode"""
    from test_plan_host import banded_matrix
    rp, ci, v = banded_matrix(6000, 1200, 8)
    if prec == 16:
        v = np.abs(v) + 0.5
    n = 6000
    st = dasp.Plan(rp, ci, v.astype(np.float64 if prec == 64 else np.float16), n, precision=prec, **kw).stats
    if not kw:
        assert st["x_window_on"] == 1 and st["n_windows_lds"] == st["n_windows"]
    if kw.get("x_window") == -1:
        assert st["x_window_on"] == 0
    check(oracle, dasp, torch_cuda, rp, ci, v, n, prec, **kw)
    # with short / long / empty rows mixed in
    rp2, ci2, v2 = util.mixed_matrix(3000, 3000, 19, values="f16" if prec == 16 else "uniform")
    check(oracle, dasp, torch_cuda, rp2, ci2, v2, 3000, prec, x_window=65536)


@pytest.mark.parametrize("prec", [64, 16])
@pytest.mark.parametrize("kw", [dict(x_window=4096, x_window_hybrid=1), dict(x_window=20000, row_window=128, x_window_hybrid=1),
                                dict(x_window=8192, x_window_hybrid=1, cid16=1, y_order=1)])
def test_hybrid_windows_mix_lds_and_global_gathers(oracle, dasp, torch_cuda, prec, kw):
    """graph-like rows (half of the columns near the row, half anywhere): the window stages its densest span only, the gathers
    outside it read global memory -- same results, same permutation; a share of the gathers strictly between 0 and 1 comes from LDS"""
    rng = np.random.default_rng(17)
    m = n = 20000
    lens = rng.choice([5, 7, 9, 14, 30, 60], size=m)
    rp = np.zeros(m + 1, np.int64)
    np.cumsum(lens, out=rp[1:])
    rows = np.repeat(np.arange(m), lens)
    near = rng.random(rp[-1]) < 0.55
    ci = np.where(near, np.clip(rows + rng.integers(-150, 151, rp[-1]), 0, n - 1), rng.integers(0, n, rp[-1])).astype(np.int32)
    v = rng.uniform(0.5, 1.5, rp[-1])
    rp = rp.astype(np.int32)
    kw = dict(kw)
    y_order = kw.pop("y_order", None)
    st = dasp.Plan(rp, ci, v.astype(np.float64 if prec == 64 else np.float16), n, precision=prec, **kw).stats
    assert st["x_window_on"] == 1 and st["x_window_hybrid"] == 1 and 0.2 < st["window_nnz_frac"] < 0.95
    check(oracle, dasp, torch_cuda, rp, ci, v, n, prec, **kw)
    auto = dasp.Plan(rp, ci, v.astype(np.float64 if prec == 64 else np.float16), n, precision=prec).stats
    assert auto["x_window_hybrid"] == 0                                     # rows of 5..60 nonzeros are not "even": auto leaves it off


def test_cop20k_standin_uses_windows(dasp, torch_cuda):
    rows, cols = dasp.synth_dims("cop20k_A", 0.25)
    rp, ci = dasp.synth_csr("cop20k_A", 0.25)
    st = dasp.Plan(rp, ci, np.ones(ci.size), cols).stats
    assert st["x_window_on"] == 1 and st["window_nnz_frac"] > 0.9



@pytest.mark.parametrize("prec", [64, 16])
@pytest.mark.parametrize("cid16", [-1, 1])
@pytest.mark.parametrize("tag,builder,m,n,seed", [("mixed", util.mixed_matrix, 3000, 2500, 7), ("wide", util.mixed_matrix, 2000, 3_000_000, 9)])
def test_cid16_modes_agree(oracle, dasp, torch_cuda, prec, cid16, tag, builder, m, n, seed):
    """16-bit column ids on / off (forced on a 3M-column matrix most chunks fall back to the 32-bit tails)"""
    rp, ci, v = builder(m, n, seed, values="f16" if prec == 16 else "uniform")
    check(oracle, dasp, torch_cuda, rp, ci, v, n, prec, cid16=cid16)
    check(oracle, dasp, torch_cuda, rp, ci, v, n, prec, cid16=cid16, x_window=100000 if n < 10000 else 0)


def test_loaded_plan_runs(oracle, dasp, torch_cuda, tmp_path):
    """a plan read back from disk gives the same y as the plan that was saved"""
    rp, ci, v = util.mixed_matrix(3000, 2500, 7)
    x = np.random.default_rng(4).uniform(-1, 1, 2500)
    plan = dasp.Plan(rp, ci, v, 2500)
    plan.save(str(tmp_path / "a.plan"))
    y0 = run_spmv(torch_cuda, plan.upload(), x, 3000, 64)
    y1 = run_spmv(torch_cuda, dasp.Plan.load(str(tmp_path / "a.plan")).upload(), x, 3000, 64)
    assert np.array_equal(y0, y1)


@pytest.mark.parametrize("prec", [64, 16])
@pytest.mark.parametrize("kw", [dict(threshold=0.01), dict(threshold=1.0), dict(block_longest=64), dict(block_longest=1000),
                                dict(long_piece=64), dict(threshold=0.5, block_longest=32, long_piece=256, x_window=40000, cid16=1)])
def test_option_corners(oracle, dasp, torch_cuda, prec, kw):
    """the reference's two tunables (threshold, block_longest: main_f64.cu:124-125) and this build's own, at their corners"""
    rp, ci, v = util.mixed_matrix(2500, 2000, 29, values="f16" if prec == 16 else "uniform")
    check(oracle, dasp, torch_cuda, rp, ci, v, 2000, prec, **kw)


def test_single_column_and_duplicates(oracle, dasp, torch_cuda):
    lens = [0, 1, 2, 3, 4, 5, 17, 64, 300, 1100]
    rp, ci, v = util.csr_from_lengths(lens, 1, 3)                   # every entry in column 0: all duplicates
    check(oracle, dasp, torch_cuda, rp, ci, v, 1, 64)
    rp, ci, v = util.csr_from_lengths([40] * 50, 3, 5)
    check(oracle, dasp, torch_cuda, rp, ci, v, 3, 16)


def test_nonfinite_values_stay_in_their_rows(dasp, torch_cuda):
    rp, ci, v = util.mixed_matrix(1500, 1200, 31)
    lens = np.diff(rp)
    bad_rows = [int(np.argmax(lens >= 256)), int(np.argmax((lens >= 5) & (lens < 256))), int(np.argmax(lens == 3))]
    v = v.copy()
    for r, val in zip(bad_rows, (np.nan, np.inf, -np.inf)):
        v[rp[r]] = val
    x = np.random.default_rng(6).uniform(0.5, 1.5, 1200)
    plan = dasp.Plan(rp, ci, v, 1200, y_order=dasp.Y_NATURAL).upload()
    y = run_spmv(torch_cuda, plan, x, 1500, 64)
    assert np.isnan(y[bad_rows[0]]) and y[bad_rows[1]] == np.inf and y[bad_rows[2]] == -np.inf
    ok = np.ones(1500, bool)
    ok[bad_rows] = False
    assert np.isfinite(y[ok]).all()


def test_bench_rank_setup_assembles_full_y(dasp, torch_cuda):
    """bench.py's multi-GPU preparation, every rank's dasp_mg plan stepped in turn on one GPU with the exchange done by hand
    (dasp_mg_product + dasp_mg_get_y_local / dasp_mg_set_x): partition by nonzeros, per-rank plans over own / other columns,
    padded slices -> three chained iterations equal to (A_s)^3 x_0"""
    import importlib.util
    import scipy.sparse as sp
    spec = importlib.util.spec_from_file_location("bench", os.path.join(ROOT, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    torch = torch_cuda
    world = 3
    rows, _ = dasp.synth_dims("HV15R", 0.02)
    rp_all, ci_all = dasp.synth_csr("HV15R", 0.02)
    lens = np.diff(rp_all)
    A = sp.csr_matrix((np.repeat(0.5 / np.maximum(lens, 1), lens), ci_all, rp_all), shape=(rows, rows))
    x0 = np.random.default_rng(4).uniform(0.5, 1.5, rows)
    for overlap in ("1", "0"):
        os.environ["DASP_BENCH_OVERLAP"] = overlap
        try:
            parts = [bench.setup_rank(torch, dasp, "HV15R", 0.02, 64, r, world) for r in range(world)]
        finally:
            del os.environ["DASP_BENCH_OVERLAP"]
        mgs = [P["mg"] for P in parts]
        nnz = [int(P["rp"][-1]) for P in parts]
        assert max(nnz) - min(nnz) <= 2 * 484                                    # balanced to within a row or two
        for mg in mgs:
            if overlap == "1":
                assert mg.overlap and mg.subplan(1) is not None and mg.nnz_local > 5 * mg.nnz_remote > 0     # banded: mostly own columns
                assert mg.subplan(0).x_len == mg.stride and mg.subplan(1).x_len == world * mg.stride
            else:
                assert not mg.overlap and mg.subplan(1) is None and mg.subplan(0).x_len == world * mg.stride
            mg.set_x(x0)
        want = x0
        for it in range(3):
            for mg in mgs:
                mg.product(0)
            full = np.concatenate([mg.get_y_local() for mg in mgs])              # what the all-gather delivers to every rank
            want = A @ want
            assert np.abs(full - want).max() <= 1e-13 * np.abs(want).max()
            for mg in mgs:
                mg.set_x(full)
        for mg in mgs:
            mg.close()


def _hv_slices(dasp, world, scale=0.02):
    import scipy.sparse as sp
    rows, _ = dasp.synth_dims("HV15R", scale)
    rp_all, ci_all = dasp.synth_csr("HV15R", scale)
    lens = np.diff(rp_all)
    val = np.repeat(0.5 / np.maximum(lens, 1), lens)
    A = sp.csr_matrix((val, ci_all, rp_all), shape=(rows, rows))
    bounds = dasp.partition_rows(rp_all, world)
    sl = []
    for r in range(world):
        r0, r1 = int(bounds[r]), int(bounds[r + 1])
        sl.append((rp_all[r0:r1 + 1] - rp_all[r0], ci_all[rp_all[r0]:rp_all[r1]], val[rp_all[r0]:rp_all[r1]]))
    return rows, A, bounds, sl


def test_fused_mg_step_three_ranks_on_one_device(dasp, torch_cuda):
    """The one-launch step (own-column workgroups, other-column workgroups, atomics into a zeroed slice, three rotating slice
    buffers) with three ranks' plans on ONE device and the test-hook exchange copying every slice into every rank's gather buffer:
    six chained iterations == (A_s)^6 x_0 from scipy, every rank holds the same gathered y, and the result is BIT-IDENTICAL to the
    two-launch form (y = own; y += other).  The in-kernel wait is exercised by the next test."""
    from dasp_amd.multi import MgPlan
    torch = torch_cuda
    world = 3
    rows, A, bounds, sl = _hv_slices(dasp, world)
    x0 = np.random.default_rng(4).uniform(0.5, 1.5, rows)
    res = {}
    for fused in (True, False):
        mgs = [MgPlan(rp, ci, v, rows, rows, bounds, r, cid16=1).upload() for r, (rp, ci, v) in enumerate(sl)]
        streams = [torch.cuda.Stream() for _ in mgs]
        for mg in mgs:
            assert mg.info["fused_step"] == 1 and mg.overlap
            if not fused:
                mg.set_fused(False)
                assert mg.info["fused_step"] == 0
            mg.set_fake_exchange(20, peers=mgs)
            mg.set_x(x0)
        want = x0
        for it in range(6):
            # the hook has no cross-rank ordering of its own (RCCL's collective has): all products, then all exchanges
            for mg, st in zip(mgs, streams):
                mg.product(st.cuda_stream)
            for mg in mgs:
                mg.check()
            for mg, st in zip(mgs, streams):
                mg.allgather(st.cuda_stream)
            torch.cuda.synchronize()
            want = A @ want
        ys = [mg.get_y() for mg in mgs]
        for y in ys[1:]:
            np.testing.assert_array_equal(y, ys[0])
        assert np.abs(ys[0] - want).max() <= 1e-13 * np.abs(want).max()
        np.testing.assert_array_equal(np.concatenate([mg.get_y_local() for mg in mgs]), ys[0])
        res[fused] = ys[0]
        for mg in mgs:
            mg.close()
    np.testing.assert_array_equal(res[True], res[False])


@pytest.mark.parametrize("fused", [True, False])
def test_push_exchange_three_ranks_in_one_process(dasp, torch_cuda, fused):
    """The direct exchange (dasp_mg_push_connect: every rank stores its slice into every rank's gather buffer and a sequence number into
    the receiver's arrived[] word; mgx.hip) with three ranks' plans in ONE process on one device, each driven from its own stream with
    plain dasp_mg_spmv calls -- the cross-rank ordering is the flags' own, unlike the copy hook's.  Chained iterations == (A_s)^n x_0
    from scipy on every rank, across a second dasp_mg_set_x after an ODD number of exchanges (both halves of the double gather buffer),
    the exchange alone (dasp_mg_allgather), and the way back: without a communicator dasp_mg_set_exchange('rccl') leaves nothing to
    exchange with."""
    from dasp_amd.multi import MgPlan
    torch = torch_cuda
    world = 3
    rows, A, bounds, sl = _hv_slices(dasp, world)
    x0 = np.random.default_rng(4).uniform(0.5, 1.5, rows)
    mgs = [MgPlan(rp, ci, v, rows, rows, bounds, r, cid16=1).upload() for r, (rp, ci, v) in enumerate(sl)]
    streams = [torch.cuda.Stream() for _ in mgs]
    blobs = [mg.push_export() for mg in mgs]
    for mg in mgs:
        if not fused:
            mg.set_fused(False)
        mg.push_connect(blobs)
        assert mg.info["exchange"] == 1 and mg.info["fused_step"] == (1 if fused else 0)
    want = x0
    for rnd, n in enumerate((5, 4)):
        for mg in mgs:
            mg.set_x(want)
        for it in range(n):
            for mg, st in zip(mgs, streams):
                mg.spmv(st.cuda_stream)
            want = A @ want
        for mg in mgs:
            mg.check()
        ys = [mg.get_y() for mg in mgs]
        for y in ys[1:]:
            np.testing.assert_array_equal(y, ys[0])
        assert np.abs(ys[0] - want).max() <= 1e-13 * np.abs(want).max()
        np.testing.assert_array_equal(np.concatenate([mg.get_y_local() for mg in mgs]), ys[0])
        want = ys[0]
    # the exchange alone: the same slices again, into the other half
    for mg, st in zip(mgs, streams):
        mg.allgather(st.cuda_stream)
    for mg in mgs:
        mg.check()
        np.testing.assert_array_equal(mg.get_y(), want)
    mgs[0].set_exchange("rccl")
    assert mgs[0].info["exchange"] == 0
    np.testing.assert_array_equal(mgs[0].get_y(), want)           # the current x moved with the switch
    with pytest.raises(dasp.DaspError, match="dasp_mg_comm_init"):
        mgs[0].spmv(streams[0].cuda_stream)
    for mg in mgs:
        mg.close()


@pytest.mark.parametrize("prec,overlap", [(16, True), (16, False), (64, False)])
def test_push_exchange_other_plan_kinds(dasp, torch_cuda, prec, overlap):
    """The direct exchange under the plans that do not run the fused step: f16 (two launches, coarse-grained gather buffer) and the
    unsplit form (ONE plan per rank reading all of x from the gather buffer, whichever half the last exchange filled): three ranks in one
    process, chained, against scipy on the same values."""
    import scipy.sparse as sp
    from dasp_amd.multi import MgPlan
    torch = torch_cuda
    world = 3
    rows, A, bounds, sl = _hv_slices(dasp, world)
    dt = np.float64 if prec == 64 else np.float16
    if prec == 16:      # values exactly representable in f16, a chain that stays in range
        sl = [(rp, ci, np.full(ci.size, 1.0 / 256.0)) for rp, ci, v in sl]
        A = sp.csr_matrix((np.full(A.nnz, 1.0 / 256.0), A.indices, A.indptr), shape=A.shape)
    x0 = np.random.default_rng(6).uniform(0.5, 1.5, rows).astype(dt)
    mgs = [MgPlan(rp, ci, v, rows, rows, bounds, r, precision=prec, overlap=overlap).upload() for r, (rp, ci, v) in enumerate(sl)]
    streams = [torch.cuda.Stream() for _ in mgs]
    blobs = [mg.push_export() for mg in mgs]
    for mg in mgs:
        mg.push_connect(blobs)
        assert mg.info["exchange"] == 1 and mg.info["fused_step"] == 0
        with pytest.raises(dasp.DaspError, match="does not qualify"):
            mg.set_fused(True)
        mg.set_x(x0)
    want = x0.astype(np.float64)
    for it in range(3):
        for mg, st in zip(mgs, streams):
            mg.spmv(st.cuda_stream)
        want = A @ want
        if prec == 16:
            want = want.astype(np.float16).astype(np.float64)       # every y is stored as f16
    for mg in mgs:
        mg.check()
    ys = [mg.get_y().astype(np.float64) for mg in mgs]
    for y in ys[1:]:
        np.testing.assert_array_equal(y, ys[0])
    tol = 1e-13 if prec == 64 else 2e-2
    assert np.abs(ys[0] - want).max() <= tol * np.abs(want).max()
    for mg in mgs:
        mg.close()


def test_reserved_stream_runs_the_step_off_32_cus(dasp, torch_cuda):
    """dasp_mg_reserved_stream: the plan's CU-masked compute stream (what RCCL as the exchange needs: its kernels do not start beside the
    product otherwise).  Three ranks in one process drive their fused steps from their reserved streams (copy-hook exchange): same bits as
    from plain streams; the handle is created once; a reserve of half the device or more is refused."""
    from dasp_amd.multi import MgPlan
    torch = torch_cuda
    world = 3
    rows, A, bounds, sl = _hv_slices(dasp, world)
    x0 = np.random.default_rng(7).uniform(0.5, 1.5, rows)
    res = {}
    for reserved in (False, True):
        mgs = [MgPlan(rp, ci, v, rows, rows, bounds, r, cid16=1).upload() for r, (rp, ci, v) in enumerate(sl)]
        if reserved:
            streams = [mg.reserved_stream(32) for mg in mgs]
            assert all(streams) and streams[0] == mgs[0].reserved_stream(32)
            assert mgs[0].reserved_stream(8) == streams[0]                      # one per plan
        else:
            keep = [torch.cuda.Stream() for _ in mgs]
            streams = [st.cuda_stream for st in keep]
        for mg in mgs:
            mg.set_fake_exchange(5, peers=mgs)
            mg.set_x(x0)
        for it in range(4):
            for mg, st in zip(mgs, streams):
                mg.product(st)
            for mg in mgs:
                mg.check()
            for mg, st in zip(mgs, streams):
                mg.allgather(st)
            torch.cuda.synchronize()
        res[reserved] = mgs[0].get_y()
        for mg in mgs:
            mg.close()
    np.testing.assert_array_equal(res[True], res[False])
    want = x0
    for it in range(4):
        want = A @ want
    assert np.abs(res[True] - want).max() <= 1e-13 * np.abs(want).max()
    mg = MgPlan(*sl[0], rows, rows, bounds, 0, cid16=1).upload()
    assert mg.reserved_stream(200) is None
    mg.close()


def test_push_exchange_reports_a_peer_that_does_not_deliver(dasp, torch_cuda, monkeypatch):
    """Direct exchange, two ranks in one process, and only rank 0 steps: its arrival kernel gives up after the time-out (5 ms here), sets
    the sticky error instead of hanging, and dasp_mg_check says whose fault it is -- without switching the exchange by itself (that is a
    collective decision).  After both ranks start again from dasp_mg_set_x the chain is right."""
    from dasp_amd.multi import MgPlan
    torch = torch_cuda
    monkeypatch.setenv("DASP_MG_TIMEOUT_MS", "5")
    world = 2
    rows, A, bounds, sl = _hv_slices(dasp, world)
    x0 = np.random.default_rng(5).uniform(0.5, 1.5, rows)
    mgs = [MgPlan(rp, ci, v, rows, rows, bounds, r, cid16=1).upload() for r, (rp, ci, v) in enumerate(sl)]
    streams = [torch.cuda.Stream() for _ in mgs]
    blobs = [mg.push_export() for mg in mgs]
    for mg in mgs:
        mg.set_fused(False)
        mg.push_connect(blobs)
        mg.set_x(x0)
    mgs[0].spmv(streams[0].cuda_stream)
    with pytest.raises(dasp.DaspError, match="did not arrive"):
        mgs[0].check()
    assert mgs[0].info["exchange"] == 1
    mgs[1].check()
    want = x0
    for mg in mgs:
        mg.set_x(x0)
    for it in range(3):
        for mg, st in zip(mgs, streams):
            mg.spmv(st.cuda_stream)
        want = A @ want
    for mg in mgs:
        mg.check()
        y = mg.get_y()
        assert np.abs(y - want).max() <= 1e-13 * np.abs(want).max()
    for mg in mgs:
        mg.close()


@pytest.mark.parametrize("fused", [1, 0])
def test_push_exchange_between_two_processes(dasp, torch_cuda, tmp_path, fused):
    """The direct exchange across PROCESSES: two ranks, each in a process of its own on this box's one GPU, map each other's gather
    buffer and flag words through hipIpcGetMemHandle / hipIpcOpenMemHandle (what eight processes on eight GPUs do) and run the chained
    dasp_mg_spmv on their own: both end with (A_s)^(5+3) x_0."""
    import subprocess
    world, iters = 2, 5
    rows, A, bounds, sl = _hv_slices(dasp, world)
    worker = os.path.join(os.path.dirname(os.path.abspath(__file__)), "_push_worker.py")
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")          # this pool's hosts support dmabuf IPC only
    procs = [subprocess.Popen([sys.executable, worker, str(tmp_path), str(r), str(world), str(iters), str(fused)], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, env=env)
             for r in range(world)]
    outs = []
    for p in procs:
        try:
            outs.append(p.communicate(timeout=600)[0])
        except subprocess.TimeoutExpired:
            for q in procs:
                q.kill()
            raise
    for p, o in zip(procs, outs):
        assert p.returncode == 0, o[-2000:]
    want = np.random.default_rng(4).uniform(0.5, 1.5, rows)
    for _ in range(iters + 3):
        want = A @ want
    ys = [np.load(tmp_path / ("%d.out.npy" % r)) for r in range(world)]
    np.testing.assert_array_equal(ys[0], ys[1])
    assert np.abs(ys[0] - want).max() <= 1e-13 * np.abs(want).max()


@pytest.mark.parametrize("seed,kw", [(31, dict(x_window=-1)), (32, dict(x_window=-1, slab_max_len=12)), (33, dict(x_window=-1, long_piece=256, block_longest=64)),
                                     (34, dict())])
def test_fused_mg_step_every_row_category(dasp, torch_cuda, seed, kw):
    """The one-launch step on a square matrix with EVERY row category in both the own-column and the other-column plan of each rank
    (empty rows, lengths 1-4 with the 1&3 pairing, medium blocks with tails, slab-stored medium rows, long rows as single pieces):
    three ranks on one device, chained, against scipy and BIT-IDENTICAL to the two-launch form.  A plan with a long row cut into
    several pieces does not qualify for the fused step and must say so (and still be right)."""
    import scipy.sparse as sp
    from dasp_amd.multi import MgPlan
    torch = torch_cuda
    m, world = 9000, 3
    rp, ci, v = util.mixed_matrix(m, m, seed, lengths=[0, 1, 2, 3, 4, 1, 3, 6, 9, 14, 27, 40, 90, 300, 700])
    lens = np.diff(rp)
    v = v / np.maximum(np.repeat(lens, lens), 1)                      # keeps the chain bounded
    A = sp.csr_matrix((v.copy(), ci.copy(), rp.copy()), shape=(m, m))       # copies: abs() below sums duplicates IN PLACE, which would re-order the rows the second form packs
    bounds = dasp.partition_rows(rp, world)
    x0 = np.random.default_rng(seed).uniform(0.5, 1.5, m)
    res = {}
    for fused in (True, False):
        mgs = []
        for r in range(world):
            r0, r1 = int(bounds[r]), int(bounds[r + 1])
            mgs.append(MgPlan(rp[r0:r1 + 1] - rp[r0], ci[rp[r0]:rp[r1]], v[rp[r0]:rp[r1]], m, m, bounds, r, cid16=1, **kw).upload())
        def qualifies(mg):      # no long row cut into several pieces, no x windows (a 9000-column x fits the LDS: auto turns them on)
            subs = [mg.subplan(w) for w in (0, 1) if mg.subplan(w) is not None]
            return all(sp_.stats["n_long_multi"] == 0 and sp_.stats["x_window_on"] == 0 for sp_ in subs)
        for mg in mgs:
            assert mg.info["fused_step"] == (1 if qualifies(mg) else 0)
            if not fused and mg.info["fused_step"]:
                mg.set_fused(False)
            mg.set_fake_exchange(5, peers=mgs)
            mg.set_x(x0)
        want = x0
        for it in range(4):
            for mg in mgs:
                mg.product(0)
            for mg in mgs:
                mg.check()
            for mg in mgs:
                mg.allgather(0)
            torch.cuda.synchronize()
            want = A @ want
        y = mgs[0].get_y()
        scale = np.abs(A) @ np.abs(x0) + 1e-300
        assert np.abs(y - want).max() <= 1e-12 * max(np.abs(want).max(), scale.max())
        for mg in mgs[1:]:
            np.testing.assert_array_equal(mg.get_y(), y)
        res[fused] = y
        for mg in mgs:
            mg.close()
    np.testing.assert_array_equal(res[True], res[False])


@pytest.mark.parametrize("name,scale,world", [("cop20k_A", 1.0, 4), ("Queen_4147", 0.01, 5), ("webbase-1M", 0.05, 2), ("nlpkkt160", 0.01, 8)])
def test_fused_mg_step_on_standins_matches_the_two_launch_form(dasp, torch_cuda, name, scale, world):
    """banded / FEM / graph stand-ins cut into `world` ranks on one device: few or all of the own-column workgroups are "marked"
    (store a row the other-column plan adds to), the dispatch-order table moves them to the front -- three chained iterations,
    the fused step BIT-IDENTICAL to the two-launch form and equal to scipy's product"""
    import scipy.sparse as sp
    from dasp_amd.multi import MgPlan
    torch = torch_cuda
    rows, cols = dasp.synth_dims(name, scale)
    rp, ci = dasp.synth_csr(name, scale)
    lens = np.diff(rp)
    v = np.random.default_rng(9).uniform(0.5, 1.5, ci.size) / np.maximum(np.repeat(lens, lens), 1)
    A = sp.csr_matrix((v.copy(), ci.copy(), rp.copy()), shape=(rows, cols))
    bounds = dasp.partition_rows(rp, world)
    x0 = np.random.default_rng(10).uniform(0.5, 1.5, cols)
    res = {}
    for fused in (True, False):
        mgs = []
        for r in range(world):
            r0, r1 = int(bounds[r]), int(bounds[r + 1])
            mgs.append(MgPlan(rp[r0:r1 + 1] - rp[r0], ci[rp[r0]:rp[r1]], v[rp[r0]:rp[r1]], rows, cols, bounds, r, cid16=1, x_window=-1, col_panels=-1).upload())
        for mg in mgs:
            assert mg.info["fused_step"] == 1
            mg.set_fused(fused)
            mg.set_fake_exchange(3, peers=mgs)
            mg.set_x(x0)
        want = x0
        for it in range(3):
            for mg in mgs:
                mg.product(0)
            for mg in mgs:
                mg.check()
            for mg in mgs:
                mg.allgather(0)
            torch.cuda.synchronize()
            want = A @ want
        res[fused] = mgs[world - 1].get_y()
        assert np.abs(res[fused] - want).max() <= 1e-12 * np.abs(want).max()
        for mg in mgs:
            mg.close()
    np.testing.assert_array_equal(res[True], res[False])


def test_fused_mg_step_without_other_column_nonzeros(dasp, torch_cuda):
    """a block-diagonal matrix: a rank's rows touch its own columns only, so there is no other-column plan and no waiting workgroup --
    the last own-column workgroup publishes "y ready" itself.  20 chained dasp_mg_spmv without host synchronisation (the exchange's
    wait kernel must see every step's flag: no time-out), result == the rank's diagonal block applied 20 times."""
    import scipy.sparse as sp
    from dasp_amd.multi import MgPlan
    m = 6000
    bounds = np.array([0, 3000, 6000], np.int32)
    rank = 1
    rp, ci, v = util.mixed_matrix(3000, 3000, 41, lengths=[0, 1, 2, 3, 4, 7, 12, 30, 64, 300])
    lens = np.diff(rp)
    v = v / np.maximum(np.repeat(lens, lens), 1)
    B = sp.csr_matrix((v.copy(), ci.copy(), rp.copy()), shape=(3000, 3000))
    mg = MgPlan(rp, ci + 3000, v, m, m, bounds, rank, cid16=1, x_window=-1).upload()       # global column ids: all inside [3000, 6000)
    assert mg.info["fused_step"] == 1 and mg.nnz_remote == 0 and mg.subplan(1) is None
    mg.set_fake_exchange(10)
    x0 = np.random.default_rng(6).uniform(0.5, 1.5, m)
    mg.set_x(x0)
    for _ in range(20):
        mg.spmv(0)
    mg.wait(0)
    mg.check()
    want = x0[3000:].copy()
    for _ in range(20):
        want = B @ want
    got = mg.get_y_local()
    assert np.abs(got - want).max() <= 1e-12 * max(np.abs(want).max(), 1e-300)
    mg.close()


def test_fused_mg_step_without_other_column_nonzeros_is_held_behind_a_slow_exchange(dasp, torch_cuda):
    """a rank WITHOUT other-column nonzeros has no waiting workgroup in its step kernel, so nothing in the kernel
    orders the caller's stream behind the exchange -- while y rotates through three slices.  With an exchange of 400 us against a product
    of a few us and no host synchronisation, step k + 3 used to overwrite the slice exchange k had not sent yet.  Every step's exchange
    copies the slice into a buffer of its own here (the hook's peer list is changed before every call), and each must hold exactly that
    step's y."""
    import scipy.sparse as sp
    from dasp_amd.multi import MgPlan
    torch = torch_cuda
    m = 6000
    bounds = np.array([0, 3000, 6000], np.int32)
    rank = 1
    rp, ci, v = util.mixed_matrix(3000, 3000, 43, lengths=[0, 1, 2, 3, 4, 7, 12, 30, 64])
    lens = np.diff(rp)
    v = v / np.maximum(np.repeat(lens, lens), 1)
    B = sp.csr_matrix((v.copy(), ci.copy(), rp.copy()), shape=(3000, 3000))
    mg = MgPlan(rp, ci + 3000, v, m, m, bounds, rank, cid16=1, x_window=-1).upload()
    assert mg.info["fused_step"] == 1 and mg.nnz_remote == 0
    steps = 9
    stride = mg.stride
    peers = [torch.zeros(2 * 2 * stride, dtype=torch.float64, device="cuda") for _ in range(steps)]      # gather-sized (two halves)
    x0 = np.random.default_rng(7).uniform(0.5, 1.5, m)
    mg.set_x(x0)
    s = torch.cuda.current_stream().cuda_stream
    for k in range(steps):
        mg.set_fake_exchange(400, [peers[k].data_ptr()])
        mg.spmv(s)
    mg.wait(s)
    mg.check()
    want = x0[3000:].copy()
    for k in range(steps):
        want = B @ want
        got = peers[k][rank * stride: rank * stride + 3000].cpu().numpy()
        assert np.abs(got - want).max() <= 1e-12 * np.abs(want).max(), "exchange %d sent another step's slice" % (k + 1)
    mg.close()


@pytest.mark.parametrize("world", [2, 3])
def test_one_stream_mg_step_ranks_in_one_process(dasp, torch_cuda, monkeypatch, world):
    """The step on ONE stream (overlap = 2, mgstep.hip dasp_mg_step2_kernel): `world` ranks of an HV15R-like partition as plans of one
    process on one device, connected through the direct exchange (in-process peers use each other's pointers), every rank on a stream
    of its own.  Each launch sends the previous slice from its head workgroups, computes the rows that read own columns at once and the
    boundary rows behind the peers' arrival flags.  Seven chained steps without host synchronisation == (A^7) x0 from scipy on every
    rank's gathered y; then dasp_mg_set_x and three more steps (the flags restart at the new epoch)."""
    import scipy.sparse as sp
    from dasp_amd.multi import MgPlan
    torch = torch_cuda
    monkeypatch.setenv("DASP_MG_SHARED_DEVICE_RANKS", str(world))
    rows, A, bounds, sl = _hv_slices(dasp, world)
    mgs = []
    for r in range(world):
        rp, ci, v = sl[r]
        mgs.append(MgPlan(rp, ci, v, rows, rows, bounds, r, cid16=1, overlap=2).upload())
    assert all(m.info["overlap"] == 1 and m.subplan(1) is None for m in mgs)
    blobs = [m.push_export() for m in mgs]
    for m in mgs:
        m.push_connect(blobs)
    assert all(m.info["fused_step"] == 2 and m.info["exchange"] == 1 for m in mgs)
    streams = [torch.cuda.Stream() for _ in range(world)]
    x0 = np.random.default_rng(11).uniform(0.5, 1.5, rows)
    want = x0.copy()
    for steps in (7, 3):
        for m in mgs:
            m.set_x(want)
        for _ in range(steps):
            for r, m in enumerate(mgs):
                m.spmv(streams[r].cuda_stream)
            want = A @ want
        for r, m in enumerate(mgs):
            m.wait(streams[r].cuda_stream)
        for m in mgs:
            m.check()
        for r, m in enumerate(mgs):
            got = m.get_y()
            assert np.abs(got - want).max() <= 1e-12 * np.abs(want).max(), (steps, r)
            r0, r1 = int(bounds[r]), int(bounds[r + 1])
            assert np.array_equal(m.get_y_local(), got[r0:r1])
    for m in mgs:
        m.close()


def test_one_stream_mg_step_rank_without_boundary_rows(dasp, torch_cuda, monkeypatch):
    """A block-diagonal matrix in the one-stream form: neither rank has a boundary row, so no workgroup of the plan waits -- ONE extra workgroup
    still does, for every peer's slice of the previous step (no rank may run two steps ahead of another: a peer's stores would meet the half
    this rank reads).  Two ranks in one process, twelve chained steps without host synchronisation == the blocks applied twelve times."""
    import scipy.sparse as sp
    from dasp_amd.multi import MgPlan
    torch = torch_cuda
    monkeypatch.setenv("DASP_MG_SHARED_DEVICE_RANKS", "2")
    m = 6000
    bounds = np.array([0, 3000, 6000], np.int32)
    blocks, mgs = [], []
    for r in range(2):
        rp, ci, v = util.mixed_matrix(3000, 3000, 51 + r, lengths=[0, 1, 2, 3, 4, 7, 12, 30, 64])
        lens = np.diff(rp)
        v = v / np.maximum(np.repeat(lens, lens), 1)
        blocks.append(sp.csr_matrix((v.copy(), ci.copy(), rp.copy()), shape=(3000, 3000)))
        mgs.append(MgPlan(rp, ci + 3000 * r, v, m, m, bounds, r, cid16=1, x_window=-1, overlap=2).upload())
    assert all(g.nnz_remote == 0 for g in mgs)
    blobs = [g.push_export() for g in mgs]
    for g in mgs:
        g.push_connect(blobs)
    assert all(g.info["fused_step"] == 2 for g in mgs)
    streams = [torch.cuda.Stream() for _ in mgs]
    x0 = np.random.default_rng(9).uniform(0.5, 1.5, m)
    for g in mgs:
        g.set_x(x0)
    for _ in range(12):
        for r, g in enumerate(mgs):
            g.spmv(streams[r].cuda_stream)
    for r, g in enumerate(mgs):
        g.wait(streams[r].cuda_stream)
    for g in mgs:
        g.check()
    want = x0.copy()
    for _ in range(12):
        want = np.concatenate([blocks[0] @ want[:3000], blocks[1] @ want[3000:]])
    for g in mgs:
        assert np.abs(g.get_y() - want).max() <= 1e-12 * np.abs(want).max()
        g.close()


def test_fused_mg_step_waits_in_the_kernel_and_times_out_cleanly(dasp, torch_cuda, monkeypatch):
    """One rank of a 2-way partition, 30 chained steps with NO host synchronisation between them and an emulated exchange of
    60 us: the other-column workgroups really wait inside the kernel for the previous exchange.  The peer's half of x never
    changes (nobody sends it), so the expected chain is y <- A_own y + A_other x0, bit-identical between the two forms.  Then a
    2 ms exchange against a 1 ms time-out: dasp_mg_check reports it, the plan drops to the two-launch form and works again."""
    import scipy.sparse as sp
    from dasp_amd.multi import MgPlan
    torch = torch_cuda
    rows, A, bounds, sl = _hv_slices(dasp, 2)
    rank = 1
    r0, r1 = int(bounds[rank]), int(bounds[rank + 1])
    rp, ci, v = sl[rank]
    Ar = sp.csr_matrix((v, ci, rp), shape=(r1 - r0, rows))
    x0 = np.random.default_rng(5).uniform(0.5, 1.5, rows)
    out = {}
    for fused in (True, False):
        mg = MgPlan(rp, ci, v, rows, rows, bounds, rank, cid16=1).upload()
        mg.set_fused(fused)
        mg.set_fake_exchange(60)
        mg.set_x(x0)
        s = torch.cuda.current_stream().cuda_stream
        for _ in range(30):
            mg.spmv(s)
        mg.wait(s)
        mg.check()
        out[fused] = mg.get_y_local()
        mg.close()
    x = x0.copy()
    for _ in range(30):
        x[r0:r1] = Ar @ x
    np.testing.assert_array_equal(out[True], out[False])
    assert np.abs(out[True] - x[r0:r1]).max() <= 1e-12 * np.abs(x[r0:r1]).max()
    monkeypatch.setenv("DASP_MG_TIMEOUT_MS", "1")
    mg = MgPlan(rp, ci, v, rows, rows, bounds, rank, cid16=1).upload()
    monkeypatch.delenv("DASP_MG_TIMEOUT_MS")
    mg.set_fake_exchange(2000)
    mg.set_x(x0)
    for _ in range(3):
        mg.spmv(0)
    with pytest.raises(dasp.DaspError) as e:
        mg.check()
    assert "timed out" in str(e.value) and mg.info["fused_step"] == 0
    mg.set_x(x0)
    mg.spmv(0)
    mg.wait(0)
    mg.check()
    x = x0.copy()
    x[r0:r1] = Ar @ x
    assert np.abs(mg.get_y_local() - x[r0:r1]).max() <= 1e-13 * np.abs(x).max()
    mg.close()


@pytest.mark.parametrize("prec", [64, 16])
@pytest.mark.parametrize("square", [True, False])
def test_mg_spmv_world_size_one_against_the_oracle(oracle, dasp, torch_cuda, prec, square):
    """dasp_mg_spmv through the C ABI with a real RCCL communicator (ncclCommInitRank / ncclAllGather called by libdasp_amd.so
    itself) at world size 1 -- all a one-GPU box allows: chained products equal the oracle's, the gathered y is the next x."""
    from dasp_amd.multi import MgPlan, unique_id
    m = 3000
    n = m if square else 2200
    dt = np.float64 if prec == 64 else np.float16
    rp, ci, v = util.mixed_matrix(m, n, 21, values="f16" if prec == 16 else "uniform", dtype=dt)
    if square:                                                        # keep x_t bounded over the chain
        v = (v.astype(np.float64) / np.maximum(np.repeat(np.diff(rp), np.diff(rp)), 1)).astype(dt)
    mg = MgPlan(rp, ci, v, m, n, np.array([0, m], np.int32), 0, precision=prec).upload()
    mg.comm_init(unique_id())
    assert mg.info["has_comm"] == 1 and mg.info["square"] == int(square)
    x = (np.random.default_rng(8).uniform(0.5, 1.5, n)).astype(dt)
    mg.set_x(x)
    want = x.astype(np.float64)
    steps = 3 if square else 1
    for _ in range(steps):
        mg.spmv(0)
        ref = oracle.csr_spmv(rp, ci, v.astype(np.float64), want)
        scale = np.maximum(oracle.csr_absrow(rp, ci, v.astype(np.float64), want), 1e-300)
        mg.wait(0)
        got = mg.get_y().astype(np.float64)
        assert (np.abs(got - ref) <= TOL[prec] * scale).all()
        np.testing.assert_array_equal(mg.get_y_local().astype(np.float64), got)
        want = got                                                    # the next product reads exactly what was gathered
    mg.close()


def test_bench_bare_multi_gpu_launch_needs_that_many_devices(torch_cuda):
    """`python bench.py --gpus 2` without a launcher on a one-GPU box: a clear error and rc 4, not a usage error"""
    import sys
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1"],
                       capture_output=True, text=True, timeout=300)
    import torch
    if torch.cuda.device_count() >= 2:
        pytest.skip("this box has two GPUs")
    assert r.returncode == 4 and "need 2 devices" in r.stderr


@pytest.mark.parametrize("tiles", [1, 2, 3, 4, 5, 7, 8, 9, 13])
@pytest.mark.parametrize("natural", [0, 1])
def test_segmented_short_tiles_go_to_a_wave_in_fours(oracle, dasp, torch_cuda, tiles, natural):
    """r5: an f64 plan with wave-segmented short rows hands kShortTpw = 4 consecutive tiles of ONE group to a wave (DevArgs::grp_wave0), the group's last wave
    fewer; no wave straddles two groups.  Every group here has `tiles` tiles (+ a partial last one), so the last wave holds 1..4 of them; exact against the CSR product
    (values and x are small integers / 8)."""
    torch = torch_cuda
    rng = np.random.default_rng(tiles)
    per_tile = {1: 64, 2: 32, 3: 20, 4: 16}                      # rows of one 64-element tile (plan.hpp short_seg_rows)
    lens = np.concatenate([np.full(per_tile[L] * (tiles - 1) + 1 + int(rng.integers(0, per_tile[L] - 1)), L) for L in (1, 2, 3, 4)] + [np.full(7, 0), np.full(33, 6)])
    lens = lens[rng.permutation(lens.size)]
    m, n = lens.size, 5000
    rp = np.zeros(m + 1, np.int32); rp[1:] = np.cumsum(lens)
    ci = rng.integers(0, n, rp[-1]).astype(np.int32)
    v = rng.integers(1, 17, rp[-1]).astype(np.float64) / 8.0
    x = rng.integers(1, 17, n).astype(np.float64) / 8.0
    for kw in (dict(), dict(col_panels=2)):
        plan = dasp.Plan(rp, ci, v, n, precision=64, short_seg=1, x_window=-1, y_order=natural, **kw)
        assert plan.stats["short_seg"] == 1
        got = run_spmv(torch, plan.upload(), x, m, 64)
        ref = oracle.csr_spmv(rp, ci, v, x)
        assert np.array_equal(got, ref if natural else ref[plan.order_rid]), kw
        plan.close()


@pytest.mark.parametrize("prec", [64, 16])
@pytest.mark.parametrize("seg", [1, -1])
@pytest.mark.parametrize("natural", [0, 1])
def test_wave_segmented_short_rows_against_the_oracle(oracle, dasp, torch_cuda, prec, seg, natural):
    """short_seg: rows of 1..4 nonzeros as one nonzero per lane with DPP row-shift sums (the north_star's short-row path) or as slabs --
    same order_rid, same counters, y == the oracle's CSR product either way.  The matrix is short rows almost only (every length 0..4
    many times over, counts that leave partial tiles in every group, 1&3 pairing active), plus a few medium / long rows."""
    torch = torch_cuda
    rng = np.random.default_rng(17)
    m, n = 50011, 30000
    lens = rng.choice([0, 1, 2, 3, 4, 9, 300], size=m, p=[0.05, 0.3, 0.2, 0.25, 0.19, 0.009, 0.001])
    rp = np.zeros(m + 1, np.int32); rp[1:] = np.cumsum(lens)
    ci = rng.integers(0, n, rp[-1]).astype(np.int32)
    dt = np.float64 if prec == 64 else np.float16
    v = rng.uniform(0.5, 1.5, rp[-1]).astype(dt)
    x = rng.uniform(0.5, 1.5, n).astype(dt)
    plan = dasp.Plan(rp, ci, v, n, precision=prec, short_seg=seg, x_window=-1, y_order=natural)
    assert plan.stats["short_seg"] == (1 if seg == 1 else 0) and plan.stats["common_13"] > 0
    P = oracle.Packed(prec, rp, ci, v.astype(np.float64), n)
    assert (plan.order_rid == P.order_rid).all()
    rows = util.decode_plan(plan)
    for slot in rng.integers(0, m, 2000):
        r = plan.order_rid[slot]
        assert rows[int(slot)][0] == ci[rp[r]:rp[r + 1]].tolist()
    got = run_spmv(torch, plan.upload(), x, m, prec)
    ref = oracle.csr_spmv(rp, ci, v.astype(np.float64), x.astype(np.float64))
    scale = oracle.csr_absrow(rp, ci, v.astype(np.float64), x.astype(np.float64))
    if not natural:
        ref, scale = ref[plan.order_rid], scale[plan.order_rid]
    assert (np.abs(got - ref) <= TOL[prec] * np.maximum(scale, 1e-300)).all()
    plan.close()


NNZ_ARRAYS = ("long_val long_cid long_cid16 long_base med_val med_cid med_cid16 med_cid8 med_base irr_val irr_cid short_val short_cid rt_val rt_cid").split()
META_ARRAYS = ("piece_ptr piece_dst piece_c16 multi_ptr multi_dst med_ptr med_c8ptr med_korig irr_ptr med_dst win_cmin win_len short_groups rt_ptr rt_start rt_mask").split()


@pytest.mark.parametrize("prec", [64, 16])
@pytest.mark.parametrize("kw", [dict(), dict(cid16=1, chunk_pairs=2), dict(cid16=-1, x_window=-1, chunk_pairs=-1), dict(x_window=100000, y_order=1), dict(x_window=-1, chunk_pairs=2, slab_max_len=4),
                                dict(part_bounds=np.array([0, 700, 2500], np.int32), part_stride=2048, y_order=1, cid16=1),
                                dict(x_window=-1, short_seg=1), dict(x_window=-1, short_seg=-1, y_order=1), dict(sort_columns=1), dict(sort_columns=1, col_panels=2, y_order=1)])
@pytest.mark.parametrize("tag,builder,m,n,seed", [("mixed", util.mixed_matrix, 3000, 2500, 7), ("pairs", util.pair_heavy_matrix, 4000, 2500, 11)])
def test_device_packed_plan_is_bit_identical(oracle, dasp, torch_cuda, prec, kw, tag, builder, m, n, seed):
    """dasp_plan_create_device (CSR on the GPU, packed by kernels) == dasp_plan_create (host packers), array by array,
    and the plan it returns is ready to run"""
    torch = torch_cuda
    dt = np.float64 if prec == 64 else np.float16
    rp, ci, v = builder(m, n, seed, values="f16" if prec == 16 else "uniform", dtype=dt)
    host = dasp.Plan(rp, ci, v, n, precision=prec, **kw)
    d_rp, d_ci, d_v = torch.from_numpy(rp).cuda(), torch.from_numpy(ci).cuda(), torch.from_numpy(v).cuda()
    dev = dasp.Plan.from_device(d_rp.data_ptr(), d_ci.data_ptr(), d_v.data_ptr(), m, n, int(rp[-1]), precision=prec, **kw)
    hs, ds = host.stats, dev.stats
    hs.pop("pre_ms"), ds.pop("pre_ms")
    assert hs == ds
    assert (host.order_rid == dev.order_rid).all()
    for name in META_ARRAYS:
        assert np.array_equal(host.host_array(name), dev.host_array(name)), name
    for name in NNZ_ARRAYS:
        h = host.host_array(name)
        d = dev.device_array(name, h.size, h.dtype)
        assert np.array_equal(h, d), name
    x = (np.random.default_rng(3).uniform(0.5, 1.5, dev.x_len)).astype(dt)
    y_h = run_spmv(torch, host.upload(), x, m, prec)
    y_d = run_spmv(torch, dev, x, m, prec)
    assert np.array_equal(y_h, y_d)


@pytest.mark.parametrize("prec", [64, 16])
@pytest.mark.parametrize("which", ["stencil", "band", "fem"])
def test_device_built_plans_take_the_same_automatic_decisions(dasp, torch_cuda, prec, which):
    """the sampled measures behind the automatic choices (row coherence -> slabs, line scatter -> x windows) run as kernels for a
    device-resident CSR: same decisions, same packed arrays as the host path"""
    import scipy.sparse as sp
    torch = torch_cuda
    dt = np.float64 if prec == 64 else np.float16
    if which == "stencil":                 # 2-D 5-point: slabs on
        nx = 120
        A = (sp.kron(sp.identity(nx), sp.diags([1, 1, 1], [-1, 0, 1], shape=(nx, nx))) +
             sp.kron(sp.diags([1, 1], [-1, 1], shape=(nx, nx)), sp.identity(nx))).tocsr()
        A.sort_indices()
        rp, ci, n = A.indptr.astype(np.int32), A.indices.astype(np.int32), nx * nx
    elif which == "band":                  # scattered inside a band: x windows on
        rp, ci = dasp.synth_csr("cop20k_A", 0.1)
        n = dasp.synth_dims("cop20k_A", 0.1)[1]
    else:                                  # runs of adjacent columns whose windows fit: x windows stay off
        rp, ci = dasp.synth_csr("HV15R", 0.01)
        n = dasp.synth_dims("HV15R", 0.01)[1]
    m = rp.size - 1
    v = np.random.default_rng(1).uniform(0.5, 1.5, ci.size).astype(dt)
    host = dasp.Plan(rp, ci, v, n, precision=prec)
    d = [torch.from_numpy(a).cuda() for a in (rp, ci, v)]
    dev = dasp.Plan.from_device(d[0].data_ptr(), d[1].data_ptr(), d[2].data_ptr(), m, n, ci.size, precision=prec)
    hs, ds = host.stats, dev.stats
    hs.pop("pre_ms"), ds.pop("pre_ms")
    assert hs == ds
    assert {"stencil": hs["n_med_blocks"] == 0 and hs["n_short_tiles"] > 0, "band": hs["x_window_on"] == 1,
            "fem": hs["x_window_on"] == 0 and hs["n_med_blocks"] > 0}[which]
    for name in META_ARRAYS:
        assert np.array_equal(host.host_array(name), dev.host_array(name)), name
    for name in NNZ_ARRAYS:
        h = host.host_array(name)
        assert np.array_equal(h, dev.device_array(name, h.size, h.dtype)), name


def test_device_packed_one_byte_ids(oracle, dasp, torch_cuda):
    """FEM-like rows, 16-bit ids forced: most chunks of the pipelined blocks are narrow (one-byte ids, moved to the front of their block).
    The device packer produces the host packer's arrays bit for bit, and the product matches the oracle"""
    torch = torch_cuda
    rp, ci = dasp.synth_csr("HV15R", 0.004)
    m, n = rp.size - 1, dasp.synth_dims("HV15R", 0.004)[1]
    v = np.random.default_rng(2).uniform(0.5, 1.5, ci.size)
    host = dasp.Plan(rp, ci, v, n, cid16=1)
    st = host.stats
    assert st["cid16_on"] == 1 and 0 < st["cid8_chunks"] < host.host_array("med_ptr")[-1]
    d = [torch.from_numpy(a).cuda() for a in (rp, ci, v)]
    dev = dasp.Plan.from_device(d[0].data_ptr(), d[1].data_ptr(), d[2].data_ptr(), m, n, ci.size, cid16=1)
    hs, ds = host.stats, dev.stats
    hs.pop("pre_ms"), ds.pop("pre_ms")
    assert hs == ds
    for name in META_ARRAYS:
        assert np.array_equal(host.host_array(name), dev.host_array(name)), name
    for name in NNZ_ARRAYS:
        h = host.host_array(name)
        assert np.array_equal(h, dev.device_array(name, h.size, h.dtype)), name
    x = np.random.default_rng(3).uniform(-1, 1, n)
    y = run_spmv(torch, dev, x, m, 64)
    order = dev.order_rid
    ref = np.array([np.dot(v[rp[r]:rp[r + 1]], x[ci[rp[r]:rp[r + 1]]]) for r in order[:2000]])
    mag = np.array([np.abs(v[rp[r]:rp[r + 1]] * x[ci[rp[r]:rp[r + 1]]]).sum() for r in order[:2000]])
    assert (np.abs(y[:2000] - ref) <= 1e-12 * np.maximum(mag, 1e-300)).all()
    assert np.array_equal(y, run_spmv(torch, host.upload(), x, m, 64))


def test_one_byte_ids_in_one_shot_blocks(oracle, dasp, torch_cuda):
    """r4: nlpkkt160-like rows of 5-28 nonzeros are ONE-SHOT blocks; in a plan whose one-shot f64 blocks are paired as a
    whole (chunk_pairs = 2: automatic beyond 1 GiB of CSR) their narrow chunks carry one-byte ids too, in whole pairs.  Host and device packer
    agree bit for bit, the product matches the oracle, and cid8 = -1 gives the same y."""
    torch = torch_cuda
    rp, ci = dasp.synth_csr("nlpkkt160", 0.01)
    m, n = rp.size - 1, dasp.synth_dims("nlpkkt160", 0.01)[1]
    v = np.random.default_rng(5).uniform(0.5, 1.5, ci.size)
    kw = dict(cid16=1, chunk_pairs=2, x_window=-1, slab_max_len=4, cid8=1)       # cid8 = 1: late r5 the automatic rule keeps one-byte ids out of one-shot blocks (they lose there)
    host = dasp.Plan(rp, ci, v, n, **kw)
    st = host.stats
    nchunks = int(host.host_array("med_ptr")[-1])
    assert dasp.Plan(rp, ci, v, n, cid16=1, chunk_pairs=2, x_window=-1, slab_max_len=4).stats["cid8_chunks"] == 0
    assert st["cid16_on"] == 1 and st["chunk_pairs"] == 2 and 0.3 * nchunks < st["cid8_chunks"] < nchunks
    d = [torch.from_numpy(a).cuda() for a in (rp, ci, v)]
    dev = dasp.Plan.from_device(d[0].data_ptr(), d[1].data_ptr(), d[2].data_ptr(), m, n, ci.size, **kw)
    hs, ds = host.stats, dev.stats
    hs.pop("pre_ms"), ds.pop("pre_ms")
    assert hs == ds
    for name in META_ARRAYS:
        assert np.array_equal(host.host_array(name), dev.host_array(name)), name
    for name in NNZ_ARRAYS:
        h = host.host_array(name)
        assert np.array_equal(h, dev.device_array(name, h.size, h.dtype)), name
    x = np.random.default_rng(6).uniform(-1, 1, n)
    y = run_spmv(torch, dev, x, m, 64)
    ref = oracle.csr_spmv(rp, ci, v, x)[dev.order_rid]
    mag = oracle.csr_absrow(rp, ci, v, x)[dev.order_rid]
    assert (np.abs(y - ref) <= 1e-12 * np.maximum(mag, 1e-300)).all()
    wide = dasp.Plan(rp, ci, v, n, **dict(kw, cid8=-1)).upload()
    assert wide.stats["cid8_chunks"] == 0
    # (the narrow chunks move to the front of their block: another order of the block's MFMA steps, so the last bits may differ)
    assert (np.abs(y - run_spmv(torch, wide, x, m, 64)) <= 1e-12 * np.maximum(mag, 1e-300)).all()
    assert np.array_equal(y, run_spmv(torch, host.upload(), x, m, 64))


def test_device_plan_rejects_bad_columns(dasp, torch_cuda):
    torch = torch_cuda
    rp = torch.tensor([0, 2, 3], dtype=torch.int32, device="cuda")
    ci = torch.tensor([0, 7, 1], dtype=torch.int32, device="cuda")      # 7 >= colA
    v = torch.ones(3, dtype=torch.float64, device="cuda")
    with pytest.raises(dasp.DaspError) as e:
        dasp.Plan.from_device(rp.data_ptr(), ci.data_ptr(), v.data_ptr(), 2, 3, 3)
    assert e.value.status == -10


def test_runs_are_bitwise_reproducible_and_linear(oracle, dasp, torch_cuda):
    """size-independent properties on a mid-size stand-in (HV15R at 1/8: 34 M nonzeros, all row categories):
    two launches give identical bits (no atomics, fixed reduction order), and A(ax + bz) = a Ax + b Az"""
    torch = torch_cuda
    rows, cols = dasp.synth_dims("HV15R", 0.125)
    rp, ci = dasp.synth_csr("HV15R", 0.125)
    rng = np.random.default_rng(11)
    v = rng.uniform(-1, 1, ci.size)
    plan = dasp.Plan(rp, ci, v, cols, y_order=dasp.Y_NATURAL).upload()
    assert plan.stats["row_long"] > 0 and plan.stats["cid16_on"] == 1
    x, z = rng.uniform(-1, 1, cols), rng.uniform(-1, 1, cols)
    y1 = run_spmv(torch, plan, x, rows, 64)
    y2 = run_spmv(torch, plan, x, rows, 64)
    assert np.array_equal(y1, y2)
    yz = run_spmv(torch, plan, z, rows, 64)
    ycomb = run_spmv(torch, plan, 0.75 * x - 1.25 * z, rows, 64)
    scale = np.maximum(oracle.csr_absrow(rp, ci, np.abs(v), np.abs(x) + np.abs(z)), 1e-300)
    assert (np.abs(ycomb - (0.75 * y1 - 1.25 * yz)) / scale).max() <= 1e-12
    # and a sample of rows against the CSR product itself
    ref = oracle.csr_spmv(rp[:20001] - rp[0], ci[: rp[20000]], v[: rp[20000]], x)
    assert (np.abs(y1[:20000] - ref) / scale[:20000]).max() <= 1e-12


def test_power_iteration_example_matches_scipy(dasp, torch_cuda, monkeypatch):
    """examples/power_iteration.py: y fed back as x for 60 steps on a symmetric stand-in == the same loop in scipy"""
    import importlib.util
    import scipy.sparse as sp
    spec = importlib.util.spec_from_file_location("power_iteration", os.path.join(ROOT, "examples", "power_iteration.py"))
    ex = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(ex)
    monkeypatch.setattr("sys.argv", ["power_iteration.py", "--workload", "Queen_4147", "--scale", "0.01", "--iters", "60"])
    lam = ex.main()
    rows, cols = dasp.synth_dims("Queen_4147", 0.01)
    rp, ci = dasp.synth_csr("Queen_4147", 0.01)
    A = sp.csr_matrix((ex.values_for(0, ci, rp), ci, rp), shape=(rows, cols))
    assert abs(A - A.T).max() == 0
    x = np.ones(cols)
    for _ in range(60):
        y = A @ x
        ref = np.linalg.norm(y)
        x = y / ref
    assert abs(lam - ref) <= 1e-10 * ref


@pytest.mark.parametrize("exchange", ["direct", "host"])
def test_bench_multi_rank_flow_on_one_gpu(torch_cuda, exchange):
    """`python bench.py --gpus 2` end to end, launched bare (bench.py spawns its own two ranks as a child torch.distributed.run
    job): partition, per-rank dasp_mg plans, exchange, max-over-ranks timing, chained + random-x checks, JSON line -- both ranks
    sharing this box's single GPU.  "direct": the bench's first choice as on a multi-GPU node -- the two processes map each other's gather
    buffers through hipIpc and exchange with stores + flag words (fused step); "host": the y slices moved through host memory."""
    import json
    import sys
    env = dict(os.environ, DASP_BENCH_SHARE_GPU="1", DASP_BENCH_EXCHANGE=exchange)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "5", "--warmup", "2",
                        "--scale", "0.02"], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    line = r.stdout.strip().splitlines()[-1]              # the driver parses the LAST line, and keeps an 8 KB tail
    assert len(line) < 4096
    out = json.loads(line)
    assert out["n_gpus"] == 2 and out["verified"] is True and out["scaling"] == "strong" and out["value"] > 0
    assert out["config"]["partition"].startswith("row ranges") and "roofline" in out and "suite" not in out
    assert out["verified_random_x"]["ok"] is True and out["verified_random_x"]["rows_checked"] > 1000
    assert out["config"]["exchange"].startswith("direct stores" if exchange == "direct" else "host memory")
    if exchange == "direct":
        assert "time-out" not in out["config"]["step_form"] and "error" not in out


def test_bench_rccl_calls_run_at_world_size_one(torch_cuda):
    """The exact sequence of the N > 1 bench (gloo control plane, unique-id broadcast, dasp_mg_comm_init -> ncclCommInitRank,
    dasp_mg_spmv -> ncclAllGather into the x layout, barrier, MAX / MIN all_reduce) on the one GPU of this box: world size 1 is
    all RCCL allows here (two ranks may not share a device), the partitioned plan + natural-order y + gathered-layout check are
    the real ones."""
    import json
    import socket
    import sys
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    env = dict(os.environ, DASP_BENCH_FORCE_DIST="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr", "127.0.0.1",
                        "--master-port", str(port), os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "5", "--warmup", "2",
                        "--scale", "0.02", "--no-suite", "--no-cpu-baseline", "--no-vendor"], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    out = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    assert out["n_gpus"] == 1 and out["verified"] is True and out["config"]["partition"].startswith("row ranges")
    assert out["config"]["exchange"] == "ncclAllGather" and out["verified_random_x"]["ok"] is True


def test_placement_trials_keep_the_bytes_and_the_results(dasp, torch_cuda, monkeypatch):
    """dasp_plan_tune_placement: the arena is copied into fresh allocations and the fastest kept -- products before and after are
    bit-identical, the packed arrays read back the same, a plan below 256 MiB is left alone.  OPT-IN since r4: dasp_plan_upload runs no
    trial unless DASP_PLACEMENT_TRIALS=n > 1 asks for it, a plan built from a device CSR only on request."""
    torch = torch_cuda
    rows, cols = dasp.synth_dims("HV15R", 0.12)
    rp, ci = dasp.synth_csr("HV15R", 0.12)
    v = np.random.default_rng(11).uniform(-1, 1, ci.size)
    x = np.random.default_rng(12).uniform(0.5, 1.5, cols)
    monkeypatch.delenv("DASP_PLACEMENT_TRIALS", raising=False)
    plan = dasp.Plan(rp, ci, v, cols, y_order=dasp.Y_NATURAL).upload()          # no trials inside the upload
    assert plan.stats["data_X"] > (256 << 20)
    y0 = run_spmv(torch, plan, x, rows, 64)
    before = plan.download_array("med_val") if hasattr(plan, "download_array") else None
    first, kept = plan.tune_placement(4)
    assert first > 0 and 0 < kept <= first
    y1 = run_spmv(torch, plan, x, rows, 64)
    np.testing.assert_array_equal(y0, y1)
    if before is not None:
        np.testing.assert_array_equal(before, plan.download_array("med_val"))
    plan.close()
    monkeypatch.setenv("DASP_PLACEMENT_TRIALS", "3")
    plan = dasp.Plan(rp, ci, v, cols, y_order=dasp.Y_NATURAL).upload()          # trials inside the upload, on request
    monkeypatch.delenv("DASP_PLACEMENT_TRIALS")
    np.testing.assert_array_equal(run_spmv(torch, plan, x, rows, 64), y0)
    plan.close()
    d_rp, d_ci, d_v = (torch.from_numpy(a).cuda() for a in (rp, ci, v))
    dplan = dasp.Plan.from_device(d_rp.data_ptr(), d_ci.data_ptr(), d_v.data_ptr(), rows, cols, ci.size, y_order=dasp.Y_NATURAL)
    np.testing.assert_array_equal(run_spmv(torch, dplan, x, rows, 64), y0)
    first, kept = dplan.tune_placement()
    assert first > 0 and kept <= first
    np.testing.assert_array_equal(run_spmv(torch, dplan, x, rows, 64), y0)
    dplan.close()
    rp2, ci2, v2 = util.mixed_matrix(3000, 2500, 5)
    small = dasp.Plan(rp2, ci2, v2, 2500).upload()
    assert small.tune_placement(3) == (0.0, 0.0)
    small.close()


@pytest.mark.parametrize("prec", [64, 16])
def test_stream_policies_give_identical_results(dasp, torch_cuda, prec):
    """plain loads (reference dasp_spmv) vs non-temporal loads (reference "bypass" dasp_spmv2): same bits, switchable at run time"""
    dt = np.float64 if prec == 64 else np.float16
    rp, ci, v = util.mixed_matrix(3000, 2500, 7, values="f16" if prec == 16 else "uniform", dtype=dt)
    x = np.random.default_rng(8).uniform(0.5, 1.5, 2500).astype(dt)
    plan = dasp.Plan(rp, ci, v, 2500, precision=prec, stream_policy=1).upload()
    y1 = run_spmv(torch_cuda, plan, x, 3000, prec)
    plan.set_stream_policy(2)
    y2 = run_spmv(torch_cuda, plan, x, 3000, prec)
    plan.set_stream_policy(0)
    y0 = run_spmv(torch_cuda, plan, x, 3000, prec)
    assert np.array_equal(y1, y2) and np.array_equal(y1, y0)
    with pytest.raises(dasp.DaspError):
        plan.set_stream_policy(5)


# ---- column panels (dasp_options_t::col_panels): P natural-order plans over column ranges + a streaming sum
@pytest.mark.parametrize("prec", [64, 16])
@pytest.mark.parametrize("kw", [dict(col_panels=2), dict(col_panels=3, cid16=1), dict(col_panels=8, x_window=-1), dict(col_panels=5, long_piece=256),
                                dict(col_panels=64), dict(col_panels=3, row_tile_max=32), dict(col_panels=2, row_tile_max=3), dict(col_panels=4, row_tile_max=-1)])
@pytest.mark.parametrize("tag,builder,m,n,seed", [("mixed", util.mixed_matrix, 3000, 2500, 7), ("pairs", util.pair_heavy_matrix, 4000, 2500, 11)])
def test_column_panels_parity(oracle, dasp, torch_cuda, prec, kw, tag, builder, m, n, seed):
    rp, ci, v = builder(m, n, seed)
    check(oracle, dasp, torch_cuda, rp, ci, v, n, prec, **kw)


@pytest.mark.parametrize("prec", [64, 16])
@pytest.mark.parametrize("m,n,lens_set", [(1000, 500, [1, 2, 3]), (130, 400, [0, 1]), (64, 128, [2]), (1, 64, [1]), (4097, 9000, [0, 5, 16, 17, 40])])
def test_column_panels_whose_rows_all_sit_in_row_tiles(oracle, dasp, torch_cuda, tmp_path, prec, m, n, lens_set):
    """row tiles at their edges: panels with NO row left for their own blocks (every row <= the bound: the panel plan is empty but for its tiles), a last tile
    of one position, bounds that cut through the lengths -- against the oracle, and again after a round trip through a plan file"""
    dt = np.float64 if prec == 64 else np.float16
    lens = np.random.default_rng(m).choice(lens_set, size=m)
    rp, ci, v = util.csr_from_lengths(lens, n, 3, values="f16" if prec == 16 else "uniform", dtype=dt)
    for T in (0, 16 if max(lens_set) > 16 else 2):
        check(oracle, dasp, torch_cuda, rp, ci, v, n, prec, col_panels=2, row_tile_max=T)
    plan = dasp.Plan(rp, ci, v, n, precision=prec, col_panels=2)
    assert plan.stats["row_tile_nnz"] == (ci.size if max(lens_set) <= 16 else plan.stats["row_tile_nnz"]) and plan.stats["n_row_tiles"] == plan.n_panels * -(-m // 64)
    path = str(tmp_path / "tiles.plan")
    plan.save(path)
    x = np.random.default_rng(5).uniform(0.5, 1.5, n).astype(dt)
    y0 = run_spmv(torch_cuda, plan.upload(), x, m, prec)
    y1 = run_spmv(torch_cuda, dasp.Plan.load(path).upload(), x, m, prec)
    assert np.array_equal(y0, y1)


@pytest.mark.parametrize("prec", [64, 16])
def test_column_panels_with_long_rows_and_unaligned_y(oracle, dasp, torch_cuda, prec):
    torch = torch_cuda
    dt = np.float64 if prec == 64 else np.float16
    m, n = 777, 40000
    lens = np.random.default_rng(5).choice([0, 1, 3, 7, 60, 500, 6000], size=m, p=[.1, .2, .2, .2, .2, .08, .02])
    rp, ci, v = util.csr_from_lengths(lens, n, 21, dtype=dt)
    check(oracle, dasp, torch, rp, ci, v, n, prec, col_panels=4)
    # y not 16-byte aligned -> the scalar form of the sum kernel; same numbers
    plan = dasp.Plan(rp, ci, v, n, precision=prec, col_panels=4).upload()
    plan.drop_host()
    x = torch.ones(n, dtype=tdtype(torch, prec), device="cuda")
    ya = torch.zeros(m + 16, dtype=x.dtype, device="cuda")
    yb = torch.zeros(m + 16, dtype=x.dtype, device="cuda")
    s = torch.cuda.current_stream().cuda_stream
    plan.spmv(x.data_ptr(), ya.data_ptr(), s)
    plan.spmv(x.data_ptr(), yb.data_ptr() + yb.element_size(), s)
    torch.cuda.synchronize()
    assert torch.equal(ya[:m], yb[1:m + 1]) and float(yb[0]) == 0 and float(yb[m + 1]) == 0
    w, e = plan.time_graph(x.data_ptr(), ya.data_ptr(), 0, 5, 20, 5)      # P + 1 kernels per SpMV captured into a graph
    assert e > 0
    plan.close()


def test_column_panels_multi_gpu_layout_and_file(oracle, dasp, torch_cuda, tmp_path):
    torch = torch_cuda
    m, n = 2000, 3000
    rp, ci, v = util.mixed_matrix(m, n, 17)
    bounds, stride = np.array([0, 1100, 3000], np.int32), 2048
    plan = dasp.Plan(rp, ci, v, n, col_panels=3, part_bounds=bounds, part_stride=stride, y_order=dasp.Y_NATURAL)
    plan.save(tmp_path / "p.plan")
    xh = np.random.default_rng(1).uniform(-1, 1, n)
    xl = np.zeros(2 * stride)
    xl[:1100] = xh[:1100]
    xl[stride:stride + 1900] = xh[1100:]
    ref = oracle.csr_spmv(rp, ci, v, xh)
    scale = np.maximum(oracle.csr_absrow(rp, ci, v, xh), 1e-300)
    for pl in (plan.upload(), dasp.Plan.load(tmp_path / "p.plan").upload()):
        got = run_spmv(torch, pl, xl, m, 64)
        assert (np.abs(got - ref) / scale).max() <= TOL[64]
        pl.close()


@pytest.mark.parametrize("name,prec,kw,form", [("powerlaw_1M", 64, {}, "panels"), ("ljournal-2008", 16, {}, "two_phase"), ("ljournal-2008", 16, {"two_phase": -1}, "panels"),
                                                ("rmat_2M", 16, {}, "two_phase"), ("rmat_2M", 16, {"two_phase": -1}, "plain")])
def test_device_csr_takes_the_same_automatic_panel_decision(dasp, torch_cuda, name, prec, kw, form):
    """the AUTOMATIC column-panel rule on a device-resident CSR (r3): the sampled rows' columns and the strided sample of column ids
    are gathered by a kernel, the rule is the host's -- same number of panels, same counters, same y, at BASELINE's full size.  r5: where the rule fires
    on an f16 matrix the plan takes the two-phase form instead (ljournal-2008; two_phase = -1 keeps the panels) -- and so does rmat_2M, whose hot columns kept it out of the panels: the two-phase rule asks only for scattered rows"""
    torch = torch_cuda
    rows, cols = dasp.synth_dims(name, 1.0)
    rp, ci = dasp.synth_csr(name, 1.0)
    dt = np.float64 if prec == 64 else np.float16
    v = np.ones(ci.size, dt)
    host = dasp.Plan(rp, ci, v, cols, precision=prec, **kw)
    d_rp, d_ci, d_v = torch.from_numpy(rp).cuda(), torch.from_numpy(ci).cuda(), torch.from_numpy(v).cuda()
    dev = dasp.Plan.from_device(d_rp.data_ptr(), d_ci.data_ptr(), d_v.data_ptr(), rows, cols, int(rp[-1]), precision=prec, **kw)
    hs, ds = host.stats, dev.stats
    hs.pop("pre_ms"), ds.pop("pre_ms")
    assert hs == ds and (hs["n_col_panels"] >= 2) == (form == "panels") and hs["two_phase"] == (form == "two_phase")
    assert (host.order_rid == dev.order_rid).all()
    x = np.ones(cols, dt)
    y_d = run_spmv(torch, dev, x, rows, prec)
    dev.close()
    y_h = run_spmv(torch, host.upload(), x, rows, prec)
    assert np.array_equal(y_h, y_d)


@pytest.mark.parametrize("prec", [64, 16])
@pytest.mark.parametrize("kw", [dict(col_panels=3), dict(col_panels=4, cid16=1, y_order=1), dict(col_panels=2, x_window=-1, long_piece=256), dict(col_panels=3, row_tile_max=7),
                                dict(col_panels=2, row_tile_max=16, y_order=1)])
def test_device_built_column_panels_are_bit_identical(oracle, dasp, torch_cuda, prec, kw):
    """r3: explicit column panels from a CSR that lives on the GPU -- the split by column range runs as two kernels (one wave per
    row, the entries of a row ranked per panel with ballots so they keep their order), every panel is packed on the device: the
    same panels, array by array, as dasp_plan_create builds on the host, and the same y.  (The AUTOMATIC panel rule still samples a
    host CSR.)"""
    torch = torch_cuda
    dt = np.float64 if prec == 64 else np.float16
    m, n = 3000, 5000
    rp, ci, v = util.mixed_matrix(m, n, 17, values="f16" if prec == 16 else "uniform", dtype=dt)
    host = dasp.Plan(rp, ci, v, n, precision=prec, **kw)
    d_rp, d_ci, d_v = torch.from_numpy(rp).cuda(), torch.from_numpy(ci).cuda(), torch.from_numpy(v).cuda()
    dev = dasp.Plan.from_device(d_rp.data_ptr(), d_ci.data_ptr(), d_v.data_ptr(), m, n, int(rp[-1]), precision=prec, **kw)
    hs, ds = host.stats, dev.stats
    hs.pop("pre_ms"), ds.pop("pre_ms")
    assert hs == ds and hs["n_col_panels"] == kw["col_panels"]
    assert (host.order_rid == dev.order_rid).all()
    for k in range(hs["n_col_panels"]):
        (hp, hb, he), (dp, db, de) = host.panel(k), dev.panel(k)
        assert (hb, he) == (db, de)
        a, b = hp.stats, dp.stats
        a.pop("pre_ms"), b.pop("pre_ms")
        assert a == b
        assert (hp.order_rid == dp.order_rid).all()
        for name in META_ARRAYS:
            assert np.array_equal(hp.host_array(name), dp.host_array(name)), (k, name)
        for name in NNZ_ARRAYS:
            h = hp.host_array(name)
            assert np.array_equal(h, dp.device_array(name, h.size, h.dtype)), (k, name)
    x = (np.random.default_rng(3).uniform(0.5, 1.5, dev.x_len)).astype(dt)
    y_h = run_spmv(torch, host.upload(), x, m, prec)
    y_d = run_spmv(torch, dev, x, m, prec)
    assert np.array_equal(y_h, y_d)
    ref = oracle.csr_spmv(rp, ci, v.astype(np.float64), x.astype(np.float64))
    scale = np.maximum(oracle.csr_absrow(rp, ci, v.astype(np.float64), x.astype(np.float64)), 1e-300)
    perm = np.arange(m) if kw.get("y_order") == 1 else dev.order_rid
    assert (np.abs(y_d - ref[perm]) <= TOL[prec] * scale[perm]).all()


@pytest.mark.parametrize("seed", range(int(os.environ.get("DASP_TEST_SEEDS", "24"))))      # DASP_TEST_SEEDS=2000 for a soak run
def test_random_option_combinations(oracle, dasp, torch_cuda, seed):
    """Feature interactions (panels x windows x 16-bit ids x partitioned x layout x piece / threshold / block_longest corners):
    every seeded combination must match the CSR product within the stated tolerance."""
    torch = torch_cuda
    rng = np.random.default_rng(1000 + seed)
    prec = int(rng.choice([64, 16]))
    dt = np.float64 if prec == 64 else np.float16
    m = int(rng.integers(1, 6000))
    n = int(rng.choice([1, 17, 300, 5000, 70000, 1_500_000]))
    kinds = [0, 1, 2, 3, 4, 5, 9, 17, 40, 130, 300, 900]
    w = rng.dirichlet(np.ones(len(kinds)) * 0.5)
    lens = np.minimum(rng.choice(kinds, size=m, p=w), n)
    shape = rng.random()
    if shape < 0.2 and n >= 64:                               # stencil-like: equal lengths, neighbouring rows read neighbouring columns
        L = int(rng.integers(5, 30))
        offs = np.sort(rng.choice(np.arange(-min(n // 2, 3000), min(n // 2, 3000)), size=L, replace=False))
        rp = (np.arange(m + 1, dtype=np.int64) * L).astype(np.int32)
        ci = ((np.arange(m, dtype=np.int64)[:, None] * max(1, n // max(m, 1)) + offs[None, :]) % n)
        ci.sort(axis=1)
        ci = ci.reshape(-1).astype(np.int32)
        v = rng.uniform(-1, 1, ci.size).astype(dt)
    elif shape < 0.45:                                        # a banded variant: windows fit
        rp, ci, v = util.csr_from_lengths(lens, min(n, 2000), int(rng.integers(1 << 30)), dtype=dt)
        ci = np.minimum(ci + (np.repeat(np.arange(m), np.diff(rp)) * max(1, (n - 2000)) // max(m, 1)).astype(np.int32), n - 1).astype(np.int32)
        for r in range(m):
            ci[rp[r]:rp[r + 1]].sort()
    else:
        rp, ci, v = util.csr_from_lengths(lens, n, int(rng.integers(1 << 30)), dtype=dt)
    kw = dict(col_panels=int(rng.choice([1, 1, 2, 3, 6])), cid16=int(rng.choice([-1, 0, 1])),
              x_window=int(rng.choice([-1, 0, 0, 40000, 163840])), row_window=int(rng.choice([0, 64, 256, 1024])), x_window_hybrid=int(rng.choice([0, 0, 1, -1])), piece_min_len=int(rng.choice([0, 0, -1, 6, 40])), chunk_pairs=int(rng.choice([0, 0, -1, 1, 2])), cid8=int(rng.choice([0, 0, -1])),
              long_piece=int(rng.choice([0, 64, 300, 4096])), block_longest=int(rng.choice([256, 256, 32, 1000])),
              threshold=float(rng.choice([0.75, 0.75, 0.3, 1.0])), y_order=int(rng.choice([0, 1])),
              slab_max_len=int(rng.choice([0, 4, 7, 16, 32])),
              row_tile_max=int(np.random.default_rng(5000 + seed).choice([0, 0, -1, 1, 5, 32])),      # (its own generator: the other draws stay what they were)
              sort_columns=int(np.random.default_rng(7000 + seed).choice([0, 0, 1])))
    part = None
    if rng.random() < 0.4 and n >= 3:
        cuts = np.sort(rng.choice(np.arange(1, n), size=min(2, n - 1), replace=False))
        bounds = np.concatenate([[0], cuts, [n]]).astype(np.int32)
        stride = int((np.diff(bounds).max() + 63) // 64 * 64)
        part = (bounds, stride)
        kw.update(part_bounds=bounds, part_stride=stride)
    xh = (rng.uniform(-1, 1, n) if prec == 64 else rng.uniform(0.5, 1.5, n)).astype(dt)
    if part is None:
        xl = xh
    else:
        xl = np.zeros((part[0].size - 1) * part[1], dt)
        for g in range(part[0].size - 1):
            xl[g * part[1]: g * part[1] + part[0][g + 1] - part[0][g]] = xh[part[0][g]:part[0][g + 1]]
    ref = oracle.csr_spmv(rp, ci, v.astype(np.float64), xh.astype(np.float64))
    scale = np.maximum(oracle.csr_absrow(rp, ci, v.astype(np.float64), xh.astype(np.float64)), 1e-300)
    plan = dasp.Plan(rp, ci, v, n, precision=prec, **kw).upload()
    got = run_spmv(torch, plan, xl, m, prec)
    perm = plan.order_rid if kw["y_order"] == 0 else np.arange(m)
    # binary16 results below 6.1e-5 are subnormal: their rounding step is the absolute 2^-24, whatever the row's magnitude
    floor = 2.0 ** -24 if prec == 16 else 0.0
    err = np.maximum(np.abs(got - ref[perm]) - floor, 0.0) / scale[perm]
    assert np.isfinite(got).all() and err.max() <= TOL[prec], (kw, prec, m, n, float(err.max()), int(err.argmax()))
    plan.close()


@pytest.mark.parametrize("prec", [64, 16])
@pytest.mark.parametrize("kw", [dict(), dict(y_order=1), dict(col_panels=3), dict(x_window=100000, row_window=128), dict(long_piece=64)])
def test_accumulate_mode_adds_into_y(oracle, dasp, torch_cuda, prec, kw):
    """dasp_plan_spmv_acc: y += A x through every store site (medium, long single / multi piece, short, empty rows, panel sum)"""
    torch = torch_cuda
    dt = np.float64 if prec == 64 else np.float16
    m, n = 2500, 3000
    lens = np.random.default_rng(8).choice([0, 1, 2, 3, 4, 9, 40, 300, 900], size=m, p=[.05, .15, .1, .15, .1, .2, .15, .07, .03])
    rp, ci, v = util.csr_from_lengths(lens, n, 77, values="f16" if prec == 16 else "uniform", dtype=dt)
    rng = np.random.default_rng(2)
    xh = rng.uniform(0.5, 1.5, n).astype(dt)
    y0 = rng.uniform(-3, 3, m).astype(dt)
    plan = dasp.Plan(rp, ci, v, n, precision=prec, **kw).upload()
    x = torch.from_numpy(xh).cuda()
    y = torch.from_numpy(y0).cuda()
    ya = y.clone()
    s = torch.cuda.current_stream().cuda_stream
    plan.spmv(x.data_ptr(), y.data_ptr(), s)                       # y  = A x
    plan.spmv(x.data_ptr(), ya.data_ptr(), s, accumulate=True)     # ya = y0 + A x
    torch.cuda.synchronize()
    ax, got = y.double().cpu().numpy(), ya.double().cpu().numpy()
    perm = plan.order_rid if kw.get("y_order", 0) == 0 else np.arange(m)
    ref = oracle.csr_spmv(rp, ci, v.astype(np.float64), xh.astype(np.float64))[perm]
    scale = np.maximum(oracle.csr_absrow(rp, ci, v.astype(np.float64), xh.astype(np.float64))[perm], 1e-300) + np.abs(y0.astype(np.float64))
    assert (np.abs(ax - ref) / scale).max() <= TOL[prec]
    assert (np.abs(got - (y0.astype(np.float64) + ref)) / scale).max() <= TOL[prec]
    if prec == 64 and "col_panels" not in kw:
        assert (got == y0 + ax).all()                               # same products, one extra add: bit-identical to y0 + (A x)
    plan.close()


@pytest.mark.parametrize("prec", [64, 16])
@pytest.mark.parametrize("shape", ["one_huge_row", "single_row", "single_column", "all_long", "tall_thin_len1"])
def test_extreme_shapes(oracle, dasp, torch_cuda, prec, shape):
    dt = np.float64 if prec == 64 else np.float16
    rng = np.random.default_rng(5)
    if shape == "one_huge_row":            # 3 M nonzeros in one row (2930 pieces -> the partial-sum stage) among ordinary rows
        lens = rng.choice([0, 1, 3, 7, 30], size=300)
        lens[137] = 3_000_000
        n = 4_000_000
    elif shape == "single_row":
        lens, n = np.array([70_000]), 100_000
    elif shape == "single_column":         # every row reads x[0]
        lens, n = rng.choice([0, 1, 1, 1], size=50_000), 1
    elif shape == "all_long":
        lens, n = rng.integers(256, 3000, size=400), 50_000
    else:
        lens, n = np.ones(300_000, np.int64), 7
    rp, ci, v = util.csr_from_lengths(lens, n, 9, values="f16" if prec == 16 else "uniform", dtype=dt)
    if prec == 16 and shape in ("one_huge_row", "single_row"):
        v = (v.astype(np.float64) / 64).astype(dt)          # keep the long sums inside binary16
    check(oracle, dasp, torch_cuda, rp, ci, v, n, prec)


@pytest.mark.parametrize("prec", [64, 16])
@pytest.mark.parametrize("kw", [dict(), dict(slab_max_len=4), dict(slab_max_len=32), dict(col_panels=2), dict(cid16=1, slab_max_len=6)])
def test_slab_stored_medium_rows(oracle, dasp, torch_cuda, prec, kw):
    """medium rows short enough to be stored as slabs (auto for stencil-like rows, forced otherwise): same results, same slots"""
    import scipy.sparse as sp
    nx = 90
    T = sp.diags([1, 1, 1], [-1, 0, 1], shape=(nx, nx))
    A = (sp.kron(sp.identity(nx), T) + sp.kron(sp.diags([1, 1], [-1, 1], shape=(nx, nx)), sp.identity(nx))).tocsr()     # 2-D 5-point
    A.sort_indices()
    rp, ci = A.indptr.astype(np.int32), A.indices.astype(np.int32)
    v = np.random.default_rng(3).uniform(0.5, 1.5, ci.size)
    if not kw:
        st = dasp.Plan(rp, ci, v, nx * nx, precision=64).stats
        assert st["row_block"] > 7000 and st["n_med_blocks"] == 0            # rows of 5 are "medium" by class, slabs by layout
    check(oracle, dasp, torch_cuda, rp, ci, v, nx * nx, prec, **kw)
    rp2, ci2, v2 = util.mixed_matrix(3000, 2500, 7)                            # random columns: auto keeps the blocks
    if not kw:
        assert dasp.Plan(rp2, ci2, v2, 2500).stats["n_med_blocks"] > 0
    check(oracle, dasp, torch_cuda, rp2, ci2, v2, 2500, prec, **kw)


def test_plans_with_different_window_sizes_coexist(oracle, dasp, torch_cuda):
    """the dynamic-LDS limit is a property of the kernel, not of a plan: a plan with 130 KiB windows must still run after a plan
    with 70 KiB windows was uploaded (and vice versa)"""
    torch = torch_cuda
    rng = np.random.default_rng(11)
    m = 6000
    plans = []
    for band in (7500, 7800, 4400):          # f64 spans of ~120, ~125, ~72 KiB: the narrowest is uploaded last
        cols = np.clip(rng.integers(-band, band + 1, size=(m, 12)) + np.arange(m)[:, None] + band, 0, m + 2 * band - 1)
        cols.sort(axis=1)
        rp = (np.arange(m + 1) * 12).astype(np.int32)
        ci = cols.reshape(-1).astype(np.int32)
        v = rng.uniform(-1, 1, ci.size)
        n = m + 2 * band
        plan = dasp.Plan(rp, ci, v, n).upload()
        assert plan.stats["x_window_on"] == 1
        plans.append((plan, rp, ci, v, n))
    assert plans[0][0].stats["lds_bytes"] > 100000 > plans[2][0].stats["lds_bytes"] > 65536 < plans[1][0].stats["lds_bytes"]
    for plan, rp, ci, v, n in plans:          # all three after the last upload
        xh = rng.uniform(-1, 1, n)
        got = run_spmv(torch, plan, xh, m, 64)
        ref = oracle.csr_spmv(rp, ci, v, xh)[plan.order_rid]
        scale = np.maximum(oracle.csr_absrow(rp, ci, v, xh)[plan.order_rid], 1e-300)
        assert (np.abs(got - ref) / scale).max() <= TOL[64]


# ---- the two-phase (gather-free) form of an f16 plan (opt.two_phase; kernels dasp_tp_expand_kernel / dasp_tp_reduce_kernel)
@pytest.mark.parametrize("tag,builder,m,n,seed,kw", [
    ("mixed", util.mixed_matrix, 3000, 2500, 7, dict()),
    ("mixed_small_blocks", util.mixed_matrix, 3000, 2500, 8, dict(tp_col_block=64, tp_row_block=16)),
    ("pairs", util.pair_heavy_matrix, 4000, 3000, 11, dict(tp_col_block=1024, tp_row_block=500)),
    ("wide", util.mixed_matrix, 900, 150000, 12, dict()),
    ("big_blocks", util.mixed_matrix, 20000, 70000, 13, dict(tp_col_block=65536, tp_row_block=8192)),
    ("tiny", util.mixed_matrix, 37, 50, 3, dict()),
    ("one_row", util.mixed_matrix, 1, 10, 5, dict()),
])
def test_two_phase_parity(oracle, dasp, torch_cuda, tag, builder, m, n, seed, kw):
    """every row category through the two kernels, both y orders, random values / x at the f16 tolerance and the exact all-ones mode"""
    rp, ci, v = builder(m, n, seed, values="f16")
    check(oracle, dasp, torch_cuda, rp, ci, v, n, 16, two_phase=1, **kw)


@pytest.mark.parametrize("lens", [[5] * 100, [255] * 33, [4] * 1000, [1] * 300 + [3] * 300, [0] * 70, [6, 0, 6, 0, 1], [5000, 256, 1023, 20000, 0, 700, 2]])
def test_two_phase_edges(oracle, dasp, torch_cuda, lens):
    rp, ci, v = util.csr_from_lengths(lens, 30011, 13, values="f16")
    check(oracle, dasp, torch_cuda, rp, ci, v, 30011, 16, two_phase=1)
    check(oracle, dasp, torch_cuda, rp, ci, v, 30011, 16, two_phase=1, tp_col_block=8, tp_row_block=1)


def test_two_phase_hybrid_hub_rows(oracle, dasp, torch_cuda):
    """r6: hub rows of a two-phase plan column-blocked (dasp_lcb_kernel<half>; dasp_lcb_reduce_kernel adds their sums into y behind phase 2),
    everything else two-phase: parity in both y orders and the all-ones mode, y += A x, replay from a graph"""
    torch = torch_cuda
    lens = [7] * 5000 + [40000, 33000, 9000, 300, 0, 2] + [1] * 100
    n = 140000                                                       # five column blocks of 32768: hub rows are those of >= 320 nonzeros
    rp, ci, v = util.csr_from_lengths(lens, n, 23, values="f16")
    st = dasp.Plan(rp, ci, v.astype(np.float16), n, precision=16, two_phase=1).stats
    assert st["two_phase"] == 1 and st["lcb_rows"] == 3 and st["lcb_col_block"] == 32768
    check(oracle, dasp, torch, rp, ci, v, n, 16, two_phase=1)
    check(oracle, dasp, torch, rp, ci, v, n, 16, two_phase=1, tp_col_block=4096, tp_row_block=64, long_cb=1)      # forced: every row of >= 256
    v16 = v.astype(np.float16)
    xh = np.random.default_rng(5).uniform(0.5, 1.5, n).astype(np.float16)
    ref = oracle.csr_spmv(rp, ci, v16.astype(np.float64), xh.astype(np.float64))
    scale = np.maximum(oracle.csr_absrow(rp, ci, v16.astype(np.float64), xh.astype(np.float64)), 1e-300)
    m = len(lens)
    plan = dasp.Plan(rp, ci, v16, n, precision=16, two_phase=1, y_order=dasp.Y_NATURAL).upload()
    x = torch.from_numpy(xh).cuda()
    y = torch.full((m,), 2.0, dtype=torch.float16, device="cuda")
    plan.spmv(x.data_ptr(), y.data_ptr(), 0, accumulate=True)
    torch.cuda.synchronize()
    assert (np.abs(y.double().cpu().numpy() - 2.0 - ref) <= 1e-2 * np.maximum(scale, 2.0)).all()
    a = run_spmv(torch, plan, xh, m, 16)
    # captured into a graph and replayed: the hub rows' sums are deterministic (no atomics on their path)
    wall, ev = plan.time_graph(x.data_ptr(), y.data_ptr(), 0, warmup=2, iters=6, batch=3)
    torch.cuda.synchronize()
    assert ev > 0 and np.array_equal(y.double().cpu().numpy()[[5000, 5001, 5002]], a[[5000, 5001, 5002]])
    plan.close()


def test_two_phase_empty_accumulate_and_unaligned_x(oracle, dasp, torch_cuda):
    torch = torch_cuda
    # no rows / no nonzeros: nothing to launch, y = 0
    plan = dasp.Plan(np.zeros(9, np.int32), np.zeros(0, np.int32), np.zeros(0, np.float16), 4, precision=16, two_phase=1).upload()
    assert (run_spmv(torch, plan, np.ones(4, np.float16), 8, 16) == 0).all()
    plan.close()
    rp, ci, v = util.mixed_matrix(5000, 4000, 21, values="f16")
    v = v.astype(np.float16)
    xh = np.random.default_rng(5).uniform(0.5, 1.5, 4000).astype(np.float16)
    ref = oracle.csr_spmv(rp, ci, v.astype(np.float64), xh.astype(np.float64))
    scale = np.maximum(oracle.csr_absrow(rp, ci, v.astype(np.float64), xh.astype(np.float64)), 1e-300)
    plan = dasp.Plan(rp, ci, v, 4000, precision=16, two_phase=1, y_order=dasp.Y_NATURAL).upload()
    # y += A x (dasp_plan_spmv_acc): one writer per position, the old value read once
    x = torch.from_numpy(xh).cuda()
    y = torch.full((5000,), 2.0, dtype=torch.float16, device="cuda")
    plan.spmv(x.data_ptr(), y.data_ptr(), 0, accumulate=True)
    torch.cuda.synchronize()
    got = y.double().cpu().numpy()
    assert (np.abs(got - 2.0 - ref) <= 1e-2 * np.maximum(scale, 2.0)).all()
    # x at an address that is not 16-byte aligned: phase 1 stages its slice with scalar loads instead
    xb = torch.zeros(4000 + 8, dtype=torch.float16, device="cuda")
    xb[3:4003] = x
    y2 = torch.full((5000,), float("nan"), dtype=torch.float16, device="cuda")
    plan.spmv(xb.data_ptr() + 6, y2.data_ptr(), 0)
    torch.cuda.synchronize()
    assert (np.abs(y2.double().cpu().numpy() - ref) <= 1e-2 * scale).all()
    # explicit zeros in the matrix are nonzeros of the CSR (0 * inf = nan as in the serial product); pads are not: they never touch x
    xi = xh.copy()
    xi[ci[rp[7]]] = np.inf
    want = oracle.csr_spmv(rp, ci, v.astype(np.float64), xi.astype(np.float64))
    got = run_spmv(torch, plan, xi, 5000, 16)
    assert np.array_equal(np.isfinite(got), np.isfinite(want))
    plan.close()


def test_two_phase_from_a_device_csr_and_a_plan_file(oracle, dasp, torch_cuda, tmp_path):
    """dasp_plan_create_device with two_phase: the CSR's nonzeros are fetched once, packed on the host, and the plan comes back uploaded with the same
    streams as the host-built plan; a plan file of the form loads, uploads and multiplies"""
    torch = torch_cuda
    rp, ci, v = util.mixed_matrix(6000, 5000, 31, values="f16")
    v = v.astype(np.float16)
    host = dasp.Plan(rp, ci, v, 5000, precision=16, two_phase=1)
    d_rp, d_ci, d_v = torch.from_numpy(rp.astype(np.int32)).cuda(), torch.from_numpy(ci.astype(np.int32)).cuda(), torch.from_numpy(v).cuda()
    dev = dasp.Plan.from_device(d_rp.data_ptr(), d_ci.data_ptr(), d_v.data_ptr(), 6000, 5000, ci.size, precision=16, two_phase=1)
    S, SEG = host.stats["tp_segments"], host.stats["tp_seg_elems"]
    assert dev.stats["two_phase"] == 1 and dev.stats["tp_segments"] == S
    for name, dt, cnt in (("tp_lcol", np.uint16, S * SEG), ("tp_lrow", np.uint16, S * SEG), ("tp_val", np.float16, S * SEG), ("tp_dst", np.int32, S)):
        assert np.array_equal(dev.device_array(name, cnt, dt), host.host_array(name)), name
    xh = np.random.default_rng(6).uniform(0.5, 1.5, 5000).astype(np.float16)
    ref = oracle.csr_spmv(rp, ci, v.astype(np.float64), xh.astype(np.float64))
    scale = np.maximum(oracle.csr_absrow(rp, ci, v.astype(np.float64), xh.astype(np.float64)), 1e-300)
    got = run_spmv(torch, dev, xh, 6000, 16)
    assert (np.abs(got - ref[dev.order_rid]) <= 1e-2 * scale[dev.order_rid]).all()
    # phase 1's stream: xs[e] = x[column of element e] for every stored element
    xs = dev.device_array("tp_xs", S * SEG, np.float16)
    dec_cols = np.full(S * SEG, -1, np.int64)
    unit, dst, lcol = host.host_array("tp_unit").reshape(-1, 3), host.host_array("tp_dst"), host.host_array("tp_lcol")
    cb = host.stats["tp_col_block"]
    for c, s0, s1 in unit.tolist():
        for s in range(s0, s1):
            dec_cols[dst[s] * SEG:(dst[s] + 1) * SEG] = c * cb + lcol[s * SEG:(s + 1) * SEG]
    live = host.host_array("tp_lrow") != 0xFFFF
    assert np.array_equal(xs[live], xh[dec_cols[live]])
    path = str(tmp_path / "tp.plan")
    host.save(path)
    again = dasp.Plan.load(path).upload()
    assert np.array_equal(run_spmv(torch, again, xh, 6000, 16), got)
    for p in (host, dev, again):
        p.close()


# ---- column-blocked long rows of a column-panel plan (opt.long_cb; dasp_lcb_kernel / dasp_lcb_reduce_kernel)
@pytest.mark.parametrize("prec", [64, 16])
@pytest.mark.parametrize("lens,n,kw", [
    ([5000, 256, 1023, 20000, 0, 700, 2, 300, 300] + [9] * 500, 70000, dict(col_panels=3, long_cb=1)),            # every row of >= 256 column-blocked, several column blocks
    ([300] * 64 + [1] * 200 + [0] * 30, 3000, dict(col_panels=2, long_cb=1)),                                     # one column block, pieces of one or two steps
    ([40000, 33000] + [4] * 3000, 140000, dict(col_panels=4, long_cb=0)),                                         # auto: two hubs hold most of the nonzeros
    ([5000, 700, 300] + [9] * 500, 70000, dict(col_panels=3, long_cb=1, block_longest=64)),
])
def test_column_blocked_long_rows_parity(oracle, dasp, torch_cuda, prec, lens, n, kw):
    rp, ci, v = util.csr_from_lengths(lens, n, 17, values="f16" if prec == 16 else "uniform")
    plan = dasp.Plan(rp, ci, v.astype(np.float64 if prec == 64 else np.float16), n, precision=prec, two_phase=-1, **kw)
    assert plan.stats["lcb_rows"] > 0 and plan.n_panels >= 2
    plan.close()
    check(oracle, dasp, torch_cuda, rp, ci, v, n, prec, two_phase=-1, **kw)


def test_column_blocked_long_rows_accumulate_device_csr_and_determinism(oracle, dasp, torch_cuda):
    torch = torch_cuda
    lens = [9000, 4100, 300] + [11] * 4000
    rp, ci, v = util.csr_from_lengths(lens, 50000, 23)
    xh = np.random.default_rng(2).uniform(-1, 1, 50000)
    ref = oracle.csr_spmv(rp, ci, v, xh)
    scale = np.maximum(oracle.csr_absrow(rp, ci, v, xh), 1e-300)
    host = dasp.Plan(rp, ci, v, 50000, col_panels=2, long_cb=1, y_order=dasp.Y_NATURAL)
    d_rp, d_ci, d_v = torch.from_numpy(rp.astype(np.int32)).cuda(), torch.from_numpy(ci.astype(np.int32)).cuda(), torch.from_numpy(v).cuda()
    dev = dasp.Plan.from_device(d_rp.data_ptr(), d_ci.data_ptr(), d_v.data_ptr(), len(lens), 50000, ci.size, col_panels=2, long_cb=1, y_order=dasp.Y_NATURAL)
    assert dev.stats["lcb_rows"] == host.stats["lcb_rows"] == 3 and dev.stats["lcb_elems"] == host.stats["lcb_elems"]
    m = len(lens)
    y_dev = run_spmv(torch, dev, xh, m, 64)
    y_host = run_spmv(torch, host.upload(), xh, m, 64)
    assert np.array_equal(y_dev, y_host) and (np.abs(y_host - ref) <= 1e-12 * scale).all()
    assert np.array_equal(run_spmv(torch, host, xh, m, 64), y_host)                   # the same bits in every run: no atomics anywhere in the path
    x = torch.from_numpy(xh).cuda()
    y = torch.full((m,), 3.0, dtype=torch.float64, device="cuda")
    host.spmv(x.data_ptr(), y.data_ptr(), 0, accumulate=True)                         # y += A x
    torch.cuda.synchronize()
    assert (np.abs(y.cpu().numpy() - 3.0 - ref) <= 1e-12 * np.maximum(scale, 3.0)).all()
    # captured into a graph and replayed, and on another stream of the caller's: the same bits
    wall, ev = host.time_graph(x.data_ptr(), y.data_ptr(), 0, warmup=2, iters=6, batch=3)
    torch.cuda.synchronize()
    assert ev > 0 and np.array_equal(y.cpu().numpy(), y_host)
    s2 = torch.cuda.Stream()
    y.fill_(float("nan"))
    torch.cuda.synchronize()
    for _ in range(3):
        host.spmv(x.data_ptr(), y.data_ptr(), s2.cuda_stream)
    s2.synchronize()
    assert np.array_equal(y.cpu().numpy(), y_host)
    host.close(); dev.close()

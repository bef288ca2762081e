"""bench.py's host-side helpers (no GPU): the row sample of verified_random_x reaches the long rows and every category, the oracle
sub-matrix product equals the full product on the sampled rows, the traffic entries are only attached to the kernel build they
were measured on, the gather roofline arithmetic."""
import importlib.util
import json
import os

import numpy as np

import util

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _bench():
    spec = importlib.util.spec_from_file_location("bench", os.path.join(ROOT, "bench.py"))
    b = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(b)
    return b


def test_sample_rows_and_oracle_rows(oracle):
    b = _bench()
    rng = np.random.default_rng(1)
    lens = np.concatenate([rng.choice([0, 1, 2, 3, 4, 7, 30], 300000), [5000, 9000, 700]])
    rp, ci, v = util.csr_from_lengths(lens, 50000, 9)
    idx = b.sample_rows(rp, 100000)
    assert idx.size >= 100000 and np.all(np.diff(idx) > 0)
    assert {300000, 300001, 300002} <= set(idx.tolist())                 # the longest rows are always checked
    for L in (0, 1, 2, 3, 4, 7, 30):
        assert (lens[idx] == L).sum() > 1000                             # every category in proportion
    x = rng.uniform(-1, 1, 50000)
    ref, scale = b.oracle_rows(oracle, rp, ci, v, x, idx)
    full = oracle.csr_spmv(rp, ci, v, x)
    assert np.array_equal(ref, full[idx]) and (scale >= np.abs(ref) - 1e-12).all()
    small = b.sample_rows(rp[:1001], 100000)
    assert small.tolist() == list(range(1000))                           # few rows: all of them


def test_random_inputs_are_seeded_and_in_range():
    b = _bench()
    v1, x1 = b.random_inputs(64, 1000, 77)
    v2, x2 = b.random_inputs(64, 1000, 77)
    assert np.array_equal(v1, v2) and np.array_equal(x1, x2) and v1.min() >= -1 and v1.max() <= 1
    h, xh = b.random_inputs(16, 1000, 77)
    assert h.dtype == np.float16 and xh.dtype == np.float16 and h.min() >= 0.5 and h.max() <= 1.5


def test_traffic_entries_follow_the_kernel_revision():
    b = _bench()
    rev = b.kernel_revision()
    ents = json.load(open(os.path.join(ROOT, "profiles", "traffic.json")))
    assert {e["workload"] for e in ents} >= {"HV15R", "cop20k_A", "nlpkkt160", "Queen_4147", "webbase-1M", "ljournal-2008", "powerlaw_1M"}
    e = ents[0]
    t = b.traffic_for(e["workload"], e["precision"], e["scale"], 1e9, e["kernel_rev"])
    assert t["traffic"] == e["traffic_bytes"] and t["traffic_over_algorithmic"] > 0
    stale = b.traffic_for(e["workload"], e["precision"], e["scale"], 1e9, "0" * 12)
    assert stale["traffic"] is None and "measured on kernel build" in stale["traffic_reason"]
    assert b.traffic_for("no-such-matrix", 64, 1.0, 1e9, rev)["traffic"] is None
    assert len(rev) == 12


def test_gather_roofline_arithmetic():
    b = _bench()
    g = b.gather_roofline(153_000_000, 1.0)                              # 153 M gathers in 1 ms = 153 G/s
    assert abs(g["achieved_Ggathers_per_s"] - 153.0) < 0.1 and abs(g["frac_of_l1_miss_queue"] - 1.0) < 0.01
    assert abs(g["peak_ta_Glines_per_s"] - 614.4) < 0.1
    assert b.algorithmic_bytes(10, 20, 100, 8) == (100 + 20 + 10) * 8 + 100 * 4 + 11 * 4       # data_origin1, main_f64.cu:143


def test_watchdog_ends_a_stuck_rank_with_a_json_error_line():
    """bench.py's N > 1 watchdog: no progress for the limit -> rank 0 prints one JSON line carrying "error" and the process ends
    with code 5 (never a re-exec)"""
    import json
    import subprocess
    import sys
    code = ("import sys, time, argparse; sys.path.insert(0, %r); import bench; "
            "a = argparse.Namespace(gpus=8, steps=3, warmup=1); d = bench.Watchdog(0, 1.0, a); d.kick('RCCL communicator up'); time.sleep(30)" % ROOT)
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=60, env=dict(os.environ, DASP_BENCH_RECORD_DIRS="/nonexistent"))
    assert r.returncode == 5
    line = json.loads(r.stdout.strip().splitlines()[-1])
    assert line["value"] is None and line["n_gpus"] == 8 and "RCCL communicator up" in line["error"]


def test_watchdog_keeps_a_complete_headline_record():
    """an extra that hangs AFTER the headline measurement (vendor comparator, a suite entry) must not cost the line: the watchdog prints
    the complete record with an "error" note"""
    import json
    import subprocess
    import sys
    code = ("import sys, time, argparse; sys.path.insert(0, %r); import bench; "
            "a = argparse.Namespace(gpus=1, steps=3, warmup=1); d = bench.Watchdog(0, 1.0, a); "
            "d.partial = {'metric': 'SpMV GFLOP/s (f64)', 'value': 1234.5, 'n_gpus': 1}; d.kick('suite entry x done'); time.sleep(30)" % ROOT)
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=60, env=dict(os.environ, DASP_BENCH_RECORD_DIRS="/nonexistent"))
    assert r.returncode == 5
    line = json.loads(r.stdout.strip().splitlines()[-1])
    assert line["value"] == 1234.5 and "suite entry x done" in line["error"]


def test_exchange_configurations_best_first():
    """bench.py N > 1: direct exchange before RCCL, the fused step before two launches; RCCL + fused only where its kernel can start while
    workgroups wait (CU-masked stream, or one rank); nothing at all -> the host-memory test hook"""
    f = _bench().exchange_configs
    assert f(True, True, True, True, 8) == [("direct", True), ("direct", False), ("RCCL", True), ("RCCL", False)]
    assert f(True, True, True, False, 8) == [("direct", True), ("direct", False), ("RCCL", False)]
    assert f(False, True, True, False, 1) == [("RCCL", True), ("RCCL", False)]
    assert f(True, False, False, False, 2) == [("direct", False)]
    assert f(False, False, True, False, 2) == [("host", False)]


def _worst_case_record(b, multi):
    """a canned full record with every optional key present and every free-text field long (what a run with fall-backs, errors and
    all eleven suite rows produces)"""
    names = ["HV15R", "cop20k_A", "nlpkkt160", "powerlaw_1M", "Queen_4147", "HV15R-unstructured", "webbase-1M", "ljournal-2008", "rmat_2M",
             "ljournal-2008-uniform", "webbase-1M-uniform"]
    sf = {"%s %s" % (n, "f64" if i < 6 else "f16"): 0.1234 for i, n in enumerate(names)}
    out = {"metric": "SpMV GFLOP/s (f64)", "value": 1236.27, "unit": "GFLOP/s", "n_gpus": 8 if multi else 1, "steps": 20, "warmup": 5,
           "ms_per_step": 0.445621, "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
           "config": {"workload": "HV15R synthetic stand-in " + "x" * 400, "generator": "g" * 600, "generator_rev": "26e32ff958e7", "rows": 2017169,
                      "cols": 2017169, "nnz": 275454726, "scale": 1.0, "partition": "p" * 400, "placement": {"note": "n" * 900}, "y_candidates": 6,
                      "exchange": "e" * 300, "step_form": "s" * 700, "rank0_nnz_own_columns": 1, "rank0_nnz_other_columns": 2},
           "roofline": {"bound": "hbm", "achieved": 7500.7, "peak": 8000.0, "unit": "GB/s", "frac": 0.9376, "traffic": 2913962942,
                        "traffic_over_algorithmic": 0.8709, "traffic_reason": "r" * 500, "kernel": "dasp_spmv_kernel<double>", "kernel_ms": 0.446063,
                        "algorithmic_bytes_per_launch": 3345800096, "frac_best_of_n_y": 0.98, "frac_separate_launches": 0.94, "frac_random_values": 0.9194, "launch_ms_median": 0.447,
                        "method": "m" * 900, "f64_share_at_or_above_0.6": 0.667, "suite_frac": sf, "suite_frac_random_values": dict(sf),
                        "suite_frac_mfma_form": {k: v for k, v in sf.items() if k.endswith("f16")}},
           "cpu_baseline": {"value": 5.343, "unit": "GFLOP/s", "cores": 1, "kind": "port", "sample": "c" * 500, "ms": 103.1, "host_cores_available": 256,
                            "build": "b" * 300},
           "verified": True, "region_event_ms_per_step": 0.445, "achieved_GBps_whole_job": 7508.2, "frac_hbm_roofline_whole_job": 0.9385,
           "preprocess_s": 0.322, "pre_ms_device_csr": 28.8,
           "verified_random_x": {"ok": True, "rows_checked": 103878, "max_rel_err": 4.2e-16, "tol": 1e-12, "inputs": "i" * 300, "error": "E" * 900},
           "rocsparse_csr": {"ms": 0.68, "gflops": 800.0}, "error": "X" * 2000,
           "suite": [{"workload": n, "verified": True, "verified_random_x": {"ok": True}, "gather_roofline": b.gather_roofline(10 ** 8, 0.4),
                      "note": "z" * 2000} for n in names[1:]] + [{"workload": "broken", "error": "y" * 3000}]}
    if multi:
        out["step_parts"] = {"allgather_alone_ms": 0.03, "other_column_product_ms": 0.01, "allgather_bytes_per_rank": 2017176, "note": "n" * 300}
        out["exchange_ms"] = {"direct": {"ms_per_step": 0.08, "fused": True, "steps": 20, "allgather_alone_ms": 0.03},
                              "RCCL": {"fused": False, "error": "q" * 1500}, "rccl_ranks": 8, "error": "w" * 1500}
    return out


def test_driver_line_is_short_and_complete(tmp_path, capsys):
    """the driver keeps an 8 KB tail of stdout and parses its LAST line -- r04's 22 KB line was lost.  Whatever the run did,
    the last stdout line is one JSON object < 4096 bytes with the contract's keys, roofline and cpu_baseline; the suite goes to bench_suite.json"""
    import json
    b = _bench()
    for multi in (False, True):
        out = _worst_case_record(b, multi)
        assert len(json.dumps(out)) > 20000                            # the full record is as fat as r04's
        line = b.driver_line(out)
        assert len(line) < b.LINE_LIMIT == 4096 and "\n" not in line
        rec = json.loads(line)
        for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data"):
            assert rec[k] == out[k], k
        assert rec["config"]["workload"].startswith("HV15R") and rec["config"]["nnz"] == 275454726 and "partition" in rec["config"]
        assert "model" not in rec["config"]
        r = rec["roofline"]
        assert (r["bound"], r["frac"], r["achieved"], r["peak"], r["unit"], r["traffic"]) == ("hbm", 0.9376, 7500.7, 8000.0, "GB/s", 2913962942)
        assert r["frac_best_of_n_y"] == 0.98 and rec["config"]["y_candidates"] == 6 and r["frac_separate_launches"] == 0.94      # `frac` is the single-y figure
        assert r["frac_random_values"] == 0.9194 and len(r["suite_frac"]) == 11 and len(r["suite_frac_random_values"]) == 11 and len(r["suite_frac_mfma_form"]) == 5
        c = rec["cpu_baseline"]
        assert (c["value"], c["cores"], c["kind"], c["unit"]) == (5.343, 1, "port", "GFLOP/s") and c["sample"]
        assert rec["suite_errors"] == ["broken"] and "suite" not in rec and rec["full_record"] == "bench_suite.json"
        if multi:
            assert rec["exchange_ms"]["rccl_ranks"] == 8 and rec["exchange_ms"]["direct"]["ms_per_step"] == 0.08
            assert rec["step_parts"]["allgather_alone_ms"] == 0.03
    # the canned record of the run the driver could not parse: its own keys survive
    logp = os.path.join(ROOT, "profiles", "r04_bench_full.json.log")
    out = json.loads(open(logp).read().strip().splitlines()[-1])
    rec = json.loads(b.driver_line(out))
    assert rec["ms_per_step"] == out["ms_per_step"] and rec["roofline"]["frac"] == out["roofline"]["frac"]
    assert rec["cpu_baseline"]["value"] == out["cpu_baseline"]["value"] and len(b.driver_line(out)) < 4096
    # emit(): full record to the side file, the short line LAST on stdout
    b.ROOT = str(tmp_path)
    b.emit(out)
    printed = capsys.readouterr().out.strip().splitlines()
    assert json.loads(printed[-1])["value"] == out["value"] and len(printed[-1]) < 4096
    side = json.load(open(tmp_path / "bench_suite.json"))
    assert len(side["suite"]) == len(out["suite"]) and side["roofline"] == out["roofline"]

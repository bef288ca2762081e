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
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=60)
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
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=60)
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

"""The CPU sanitizer builds of the host side (dasp_amd/csrc/Makefile: `make san`, `make tsan`; profiles/r04_sanitizers.md).  Each takes 1-2 minutes on 8 cores,
so they run only when asked: DASP_RUN_SANITIZERS=1 python -m pytest tests/test_sanitizers.py"""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.skipif(os.environ.get("DASP_RUN_SANITIZERS") != "1", reason="set DASP_RUN_SANITIZERS=1 (each build + run takes 1-2 minutes)")
@pytest.mark.parametrize("target", ["san", "tsan"])
def test_host_side_under_sanitizers(target):
    r = subprocess.run(["make", "-C", os.path.join(ROOT, "dasp_amd", "csrc"), target], capture_output=True, text=True, timeout=1800)
    out = r.stdout + r.stderr
    assert r.returncode == 0, out[-3000:]
    assert "0 failed checks" in out and "ERROR: " not in out and "WARNING: ThreadSanitizer" not in out, out[-3000:]

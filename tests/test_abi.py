"""The C-ABI library loads and exports every symbol include/dasp_amd.h declares (no compute)."""
import ctypes as C
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    text = open(os.path.join(ROOT, "include", "dasp_amd.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(dasp_[a-z0-9_]+)\s*\(", text)))


def test_header_symbols_exported(dasp):
    L = C.CDLL(dasp.SO_PATH)
    syms = declared_symbols()
    assert len(syms) >= 20
    for s in syms:
        assert hasattr(L, s), "libdasp_amd.so does not export " + s
    assert sorted(dasp._lib.EXPORTS) == syms


def test_version_and_defaults(dasp):
    L = dasp._lib.lib()
    assert L.dasp_version().decode().startswith("dasp_amd")
    o = dasp._lib.Options()
    L.dasp_options_default(C.byref(o))
    assert (o.threshold, o.block_longest, o.y_order) == (0.75, 256, 0)   # main_f64.cu:124-125


def test_argument_errors(dasp):
    import numpy as np
    with pytest.raises(dasp.DaspError) as e:
        dasp.Plan(np.array([0, 2], np.int32), np.array([0, 9], np.int32), np.ones(2), colA=3)
    assert e.value.status == -10 and "column" in str(e.value)
    with pytest.raises(dasp.DaspError):
        dasp.Plan(np.array([0, 2, 1], np.int32), np.array([0], np.int32), np.ones(1), colA=3)
    with pytest.raises(dasp.DaspError) as e:
        dasp.mmio_allinone("/nonexistent/x.mtx")
    assert e.value.status == -1

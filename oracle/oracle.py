"""ctypes front-end of the CPU oracle (oracle/dasp_oracle.c).

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and the
cpu_baseline leg of bench.py.  Nothing under dasp_amd/ imports this module.
"parity unpinned" -- see oracle/dasp_oracle.h for what is and is not pinned.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None
_REF = None

_INT_FIELDS = (
    "precision rowA colA nnzA row_long row_block row_zero rowloop short_row_1 short_row_2 "
    "short_row_3 short_row_4 common_13 short_row_34 nnz_short nnz_long origin_nnz_reg nnz_irreg "
    "fill0_nnz_short13 fill0_nnz_short34 fill0_nnz_short22 fill0_nnz_short fill0_nnz_long "
    "fill0_nnz_reg threadblock13 threadblock34 threadblock22 blocknum warp_number BlockNum_long "
    "offset_short1"
).split()
_ARR_FIELDS = {
    "order_rid": np.int32, "short_val": np.float64, "short_cid": np.int32, "long_val": np.float64,
    "long_cid": np.int32, "long_rpt_new": np.int32, "reg_val": np.float64, "reg_cid": np.int32,
    "block_ptr": np.int32, "irreg_val": np.float64, "irreg_cid": np.int32, "irreg_rpt": np.int32,
}


def build(force=False):
    """(Re)build liboracle.so and, when /root/reference exists, oracle/_ref."""
    so = os.path.join(_HERE, "liboracle.so")
    src = os.path.join(_HERE, "dasp_oracle.c")
    if force or not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "-s", "all"])
    return so


def lib():
    global _LIB
    if _LIB is None:
        L = C.CDLL(build())
        ip, dp = C.POINTER(C.c_int), C.POINTER(C.c_double)
        L.oracle_mm_read_banner.argtypes = [C.c_char_p, C.c_char_p]
        L.oracle_mm_read_size.argtypes = [C.c_char_p, ip, ip, ip]
        L.oracle_mmio_allinone.argtypes = [C.c_char_p, ip, ip, ip, ip, C.POINTER(ip), C.POINTER(ip), C.POINTER(dp)]
        L.oracle_free.argtypes = [C.c_void_p]
        L.oracle_mm_read_coo.argtypes = [C.c_char_p, C.c_char_p, ip, ip, ip, C.POINTER(ip), C.POINTER(ip), C.POINTER(dp), C.POINTER(dp)]
        L.oracle_exclusive_scan.argtypes = [C.c_void_p, C.c_int]
        L.oracle_radix_sort_desc.argtypes = [C.c_void_p, C.c_void_p, C.c_int]
        L.oracle_csr_spmv_f64.argtypes = [C.c_int] + [C.c_void_p] * 5
        L.oracle_csr_absrow_f64.argtypes = [C.c_int] + [C.c_void_p] * 5
        L.oracle_round_f16.argtypes = [C.c_double]
        L.oracle_round_f16.restype = C.c_double
        L.oracle_dasp_new.restype = C.c_void_p
        L.oracle_dasp_pack.argtypes = [C.c_int] * 4 + [C.c_void_p] * 3 + [C.c_double, C.c_int, C.c_void_p]
        L.oracle_dasp_free.argtypes = [C.c_void_p]
        L.oracle_dasp_eval.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
        L.oracle_dasp_int.argtypes = [C.c_void_p, C.c_char_p]
        L.oracle_dasp_data_X.argtypes = [C.c_void_p]; L.oracle_dasp_data_X.restype = C.c_longlong
        L.oracle_dasp_rate_fill0.argtypes = [C.c_void_p]; L.oracle_dasp_rate_fill0.restype = C.c_double
        L.oracle_dasp_arr.argtypes = [C.c_void_p, C.c_char_p, ip]
        L.oracle_dasp_arr.restype = C.c_void_p
        L.oracle_fnv1a_i32.argtypes = [C.c_void_p, C.c_longlong]
        L.oracle_fnv1a_i32.restype = C.c_ulonglong
        _LIB = L
    return _LIB


def ref_mmio():
    """The reference's own mmio.h built by oracle/Makefile (None when never built)."""
    global _REF
    if _REF is None:
        so = os.path.join(_HERE, "_ref", "libref_mmio.so")
        if not os.path.exists(so):
            return None
        R = C.CDLL(so)
        ip = C.POINTER(C.c_int)
        R.ref_mm_read_banner_path.argtypes = [C.c_char_p, C.c_char_p]
        R.ref_mm_read_size_path.argtypes = [C.c_char_p, ip, ip, ip]
        R.ref_mm_read_crd_data_path.argtypes = [C.c_char_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, ip]
        R.ref_mm_read_crd_entries_path.argtypes = [C.c_char_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, ip]
        _REF = R
    return _REF


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


def mm_read_banner(path):
    tc = C.create_string_buffer(4)
    rc = lib().oracle_mm_read_banner(os.fsencode(path), tc)
    return rc, tc.raw.decode("latin1")


def mm_read_size(path):
    M, N, nz = C.c_int(), C.c_int(), C.c_int()
    rc = lib().oracle_mm_read_size(os.fsencode(path), C.byref(M), C.byref(N), C.byref(nz))
    return rc, M.value, N.value, nz.value


def mmio_allinone(path):
    """-> (rc, m, n, nnz, is_symmetric, row_ptr, col_idx, val[f64])"""
    L = lib()
    ip, dp = C.POINTER(C.c_int), C.POINTER(C.c_double)
    m, n, nnz, sym = C.c_int(), C.c_int(), C.c_int(), C.c_int()
    rp, ci, v = ip(), ip(), dp()
    rc = L.oracle_mmio_allinone(os.fsencode(path), C.byref(m), C.byref(n), C.byref(nnz), C.byref(sym),
                                C.byref(rp), C.byref(ci), C.byref(v))
    if rc != 0:
        return rc, 0, 0, 0, 0, None, None, None
    row_ptr = np.ctypeslib.as_array(rp, (m.value + 1,)).copy()
    col_idx = np.ctypeslib.as_array(ci, (max(nnz.value, 1),))[: nnz.value].copy()
    val = np.ctypeslib.as_array(v, (max(nnz.value, 1),))[: nnz.value].copy()
    for q in (rp, ci, v):
        L.oracle_free(C.cast(q, C.c_void_p))
    return rc, m.value, n.value, nnz.value, sym.value, row_ptr, col_idx, val


def mm_read_coo(path):
    """The entry loop of mmio_allinone alone -> (rc, typecode, M, N, I, J, re, im): file-order COO, 0-based."""
    L = lib()
    ip, dp = C.POINTER(C.c_int), C.POINTER(C.c_double)
    tc = C.create_string_buffer(4)
    M, N, nz = C.c_int(), C.c_int(), C.c_int()
    I, J, re, im = ip(), ip(), dp(), dp()
    rc = L.oracle_mm_read_coo(os.fsencode(path), tc, C.byref(M), C.byref(N), C.byref(nz), C.byref(I), C.byref(J), C.byref(re), C.byref(im))
    if rc != 0:
        return rc, tc.raw.decode("latin1"), 0, 0, None, None, None, None
    k = nz.value
    arr = lambda p: np.ctypeslib.as_array(p, (max(k, 1),))[:k].copy()
    out = (rc, tc.raw.decode("latin1"), M.value, N.value, arr(I), arr(J), arr(re), arr(im))
    for q in (I, J, re, im):
        L.oracle_free(C.cast(q, C.c_void_p))
    return out


def ref_read_crd(path, cap=1 << 22):
    """The REFERENCE's entry parsers on `path` (oracle/_ref, built from src/mmio.h:866-980):
    -> dict(bulk=(rc, I, J, val), entries=(rc, I, J, re, im), nz) with I, J 1-based as the reference returns them."""
    R = ref_mmio()
    if R is None:
        return None
    I, J = np.zeros(cap, np.int32), np.zeros(cap, np.int32)
    val = np.full(2 * cap, np.nan)
    nz = C.c_int(0)
    rc = R.ref_mm_read_crd_data_path(os.fsencode(path), cap, _p(I), _p(J), _p(val), C.byref(nz))
    k = max(nz.value, 0)
    bulk = (rc, I[:k].copy(), J[:k].copy(), val[:2 * k].copy())
    I2, J2 = np.zeros(cap, np.int32), np.zeros(cap, np.int32)
    re, im = np.zeros(cap), np.zeros(cap)
    rc2 = R.ref_mm_read_crd_entries_path(os.fsencode(path), cap, _p(I2), _p(J2), _p(re), _p(im), C.byref(nz))
    return dict(bulk=bulk, entries=(rc2, I2[:k].copy(), J2[:k].copy(), re[:k].copy(), im[:k].copy()), nz=k)


def exclusive_scan(a):
    a = np.ascontiguousarray(a, dtype=np.int32).copy()
    lib().oracle_exclusive_scan(_p(a), a.size)
    return a


def radix_sort_desc(key, idx):
    key = np.ascontiguousarray(key, dtype=np.int32).copy()
    idx = np.ascontiguousarray(idx, dtype=np.int32).copy()
    lib().oracle_radix_sort_desc(_p(key), _p(idx), key.size)
    return key, idx


def csr_spmv(row_ptr, col_idx, val, x):
    m = row_ptr.size - 1
    row_ptr = np.ascontiguousarray(row_ptr, np.int32)
    col_idx = np.ascontiguousarray(col_idx, np.int32)
    val = np.ascontiguousarray(val, np.float64)
    x = np.ascontiguousarray(x, np.float64)
    y = np.empty(m, np.float64)
    lib().oracle_csr_spmv_f64(m, _p(row_ptr), _p(col_idx), _p(val), _p(x), _p(y))
    return y


_NATIVE = None


def native_csr_spmv():
    """The serial CSR loop compiled ON THIS HOST with -O3 -march=native (BASELINE.md section 3 asks the CPU baseline for both):
    returns (callable like csr_spmv, build description).  Falls back to the portable liboracle.so when no compiler is around."""
    global _NATIVE
    if _NATIVE is None:
        import tempfile
        so = os.path.join(tempfile.gettempdir(), "liboracle_native_%d.so" % os.getuid())
        src = os.path.join(_HERE, "dasp_oracle.c")
        what = "gcc -O3 -march=native, built on this host"
        try:
            subprocess.check_call([os.environ.get("CC", "gcc"), "-O3", "-march=native", "-fPIC", "-std=c11", "-w", "-shared", "-o", so, src, "-lm"],
                                  stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
            L = C.CDLL(so)
            L.oracle_csr_spmv_f64.argtypes = [C.c_int] + [C.c_void_p] * 5
        except Exception:
            L, what = lib(), "portable liboracle.so (-O3 -march=x86-64-v2): no compiler on this host"
        _NATIVE = (L, what)
    L, what = _NATIVE

    def run(row_ptr, col_idx, val, x):
        m = row_ptr.size - 1
        y = np.empty(m, np.float64)
        L.oracle_csr_spmv_f64(m, _p(row_ptr), _p(col_idx), _p(val), _p(x), _p(y))
        return y
    return run, what


def csr_absrow(row_ptr, col_idx, val, x):
    m = row_ptr.size - 1
    row_ptr = np.ascontiguousarray(row_ptr, np.int32)
    col_idx = np.ascontiguousarray(col_idx, np.int32)
    val = np.ascontiguousarray(val, np.float64)
    x = np.ascontiguousarray(x, np.float64)
    s = np.empty(m, np.float64)
    lib().oracle_csr_absrow_f64(m, _p(row_ptr), _p(col_idx), _p(val), _p(x), _p(s))
    return s


def round_f16(a):
    return np.asarray(a, np.float64).astype(np.float16).astype(np.float64)


class Packed:
    """Result of oracle_dasp_pack: reference-geometry DASP arrays as numpy copies."""

    def __init__(self, precision, row_ptr, col_idx, val, n_cols, threshold=0.75, block_longest=256):
        L = lib()
        self._rp = np.ascontiguousarray(row_ptr, np.int32)
        self._ci = np.ascontiguousarray(col_idx, np.int32)
        self._v = np.ascontiguousarray(val, np.float64)
        m = self._rp.size - 1
        self._h = L.oracle_dasp_new()
        rc = L.oracle_dasp_pack(precision, m, n_cols, int(self._ci.size), _p(self._rp), _p(self._ci), _p(self._v),
                                threshold, block_longest, self._h)
        if rc != 0:
            raise ValueError("oracle_dasp_pack rc=%d" % rc)
        for f in _INT_FIELDS:
            setattr(self, f, L.oracle_dasp_int(self._h, f.encode()))
        self.data_X = int(L.oracle_dasp_data_X(self._h))
        self.rate_fill0 = float(L.oracle_dasp_rate_fill0(self._h))
        for f, dt in _ARR_FIELDS.items():
            n = C.c_int()
            ptr = L.oracle_dasp_arr(self._h, f.encode(), C.byref(n))
            if n.value > 0:
                ct = C.c_double if dt == np.float64 else C.c_int
                arr = np.ctypeslib.as_array(C.cast(ptr, C.POINTER(ct)), (n.value,)).copy()
            else:
                arr = np.zeros(0, dt)
            setattr(self, f, arr)

    def eval(self, x):
        x = np.ascontiguousarray(x, np.float64)
        y = np.empty(self.rowA, np.float64)
        lib().oracle_dasp_eval(self._h, _p(x), _p(y))
        return y

    def __del__(self):
        try:
            if self._h:
                lib().oracle_dasp_free(self._h)
                lib().oracle_free(self._h)
                self._h = None
        except Exception:
            pass


def fnv1a_i32(a):
    a = np.ascontiguousarray(a, np.int32)
    return int(lib().oracle_fnv1a_i32(_p(a), a.size))

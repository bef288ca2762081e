"""Host-side mirror of the reference's interface for the DASP SpMV path, on top of the C ABI.

Names and argument meaning follow the reference (paths relative to the reference tree):
  mmio_allinone(filename)            src/mmio_highlevel.h:608-610
  spmv_all(filename, csrValA, csrRowPtrA, csrColIdxA, X_val, rowA, colA, nnzA, NUM, threshold,
           block_longest)            src/dasp_f64.h:486-487 / src/dasp_f16.h:1015-1016
  Plan                               the same work split into create / upload / spmv
numpy carries host arrays; torch (if used by the caller) only supplies device pointers and
streams.  Half precision crosses as numpy.float16.
"""
import ctypes as C
import os

import numpy as np

from . import _lib
from ._lib import DaspError, Options, Stats

Y_PERMUTED, Y_NATURAL = 0, 1


def _vp(a):
    return a.ctypes.data_as(C.c_void_p)


def _dtype(precision):
    if precision == 64:
        return np.float64
    if precision == 16:
        return np.float16
    raise ValueError("precision must be 64 or 16")


def mmio_allinone(filename, precision=64):
    """MatrixMarket -> CSR.  Returns (m, n, nnz, isSymmetric, csrRowPtr, csrColIdx, csrVal).
    Raises DaspError with the reference's codes: -1 open, -2 banner, -4 size line (-5 bad entry)."""
    L = _lib.lib()
    ip = C.POINTER(C.c_int)
    m, n, nnz, sym = C.c_int(), C.c_int(), C.c_int(), C.c_int()
    rp, ci = ip(), ip()
    if precision == 64:
        v = C.POINTER(C.c_double)()
        rc = L.dasp_mmio_allinone_f64(C.byref(m), C.byref(n), C.byref(nnz), C.byref(sym), C.byref(rp), C.byref(ci), C.byref(v), os.fsencode(filename))
    else:
        v = C.POINTER(C.c_uint16)()
        rc = L.dasp_mmio_allinone_f16(C.byref(m), C.byref(n), C.byref(nnz), C.byref(sym), C.byref(rp), C.byref(ci), C.byref(v), os.fsencode(filename))
    _lib.check(rc)
    k = nnz.value
    row_ptr = np.ctypeslib.as_array(rp, (m.value + 1,)).copy()
    col_idx = np.ctypeslib.as_array(ci, (max(k, 1),))[:k].copy()
    raw = np.ctypeslib.as_array(v, (max(k, 1),))[:k].copy()
    val = raw if precision == 64 else raw.view(np.float16)
    for q in (rp, ci, v):
        L.dasp_free(C.cast(q, C.c_void_p))
    return m.value, n.value, k, sym.value, row_ptr, col_idx, val


def csr_save(path, row_ptr, col_idx, val, n_cols, is_symmetric=0, precision=64):
    """Binary cache of a loaded CSR (dasp_csr_save)."""
    rp = np.ascontiguousarray(row_ptr, np.int32)
    ci = np.ascontiguousarray(col_idx, np.int32)
    v = np.ascontiguousarray(val, _dtype(precision))
    _lib.check(_lib.lib().dasp_csr_save(os.fsencode(path), precision, rp.size - 1, int(n_cols), int(ci.size), int(is_symmetric), _vp(rp), _vp(ci), _vp(v)))


def csr_load(path, precision=64):
    """-> (m, n, nnz, isSymmetric, csrRowPtr, csrColIdx, csrVal), as mmio_allinone returned them."""
    L = _lib.lib()
    ip = C.POINTER(C.c_int)
    m, n, nnz, sym = C.c_int(), C.c_int(), C.c_int(), C.c_int()
    rp, ci, v = ip(), ip(), C.c_void_p()
    _lib.check(L.dasp_csr_load(os.fsencode(path), precision, C.byref(m), C.byref(n), C.byref(nnz), C.byref(sym), C.byref(rp), C.byref(ci), C.byref(v)))
    k = nnz.value
    row_ptr = np.ctypeslib.as_array(rp, (m.value + 1,)).copy()
    col_idx = np.ctypeslib.as_array(ci, (max(k, 1),))[:k].copy()
    ct = C.c_double if precision == 64 else C.c_uint16
    raw = np.ctypeslib.as_array(C.cast(v, C.POINTER(ct)), (max(k, 1),))[:k].copy()
    val = raw if precision == 64 else raw.view(np.float16)
    for q in (C.cast(rp, C.c_void_p), C.cast(ci, C.c_void_p), v):
        L.dasp_free(q)
    return m.value, n.value, k, sym.value, row_ptr, col_idx, val


class Plan:
    """DASP plan: classifier + packers on the host, kernels on the current HIP device."""

    def __init__(self, csrRowPtr, csrColIdx, csrVal, colA, precision=64, threshold=0.75, block_longest=256,
                 y_order=Y_PERMUTED, long_piece=0, host_threads=0, part_bounds=None, part_stride=0, x_window=0, row_window=0, cid16=0, stream_policy=0,
                 col_panels=0, slab_max_len=0, x_window_hybrid=0, piece_min_len=0, chunk_pairs=0, cid8=0, short_seg=0, row_tile_max=0, sort_columns=0,
                 two_phase=0, tp_col_block=0, tp_row_block=0, long_cb=0):
        L = _lib.lib()
        self.precision = precision
        dt = _dtype(precision)
        rp = np.ascontiguousarray(csrRowPtr, np.int32)
        ci = np.ascontiguousarray(csrColIdx, np.int32)
        v = np.ascontiguousarray(csrVal, dt)
        self.rowA, self.colA, self.nnzA = rp.size - 1, int(colA), int(ci.size)
        opt = Options()
        L.dasp_options_default(C.byref(opt))
        opt.threshold, opt.block_longest, opt.y_order = threshold, block_longest, y_order
        opt.long_piece, opt.host_threads = long_piece, host_threads
        opt.x_window, opt.row_window, opt.cid16, opt.stream_policy = x_window, row_window, cid16, stream_policy
        opt.col_panels, opt.slab_max_len, opt.x_window_hybrid, opt.piece_min_len = col_panels, slab_max_len, x_window_hybrid, piece_min_len
        opt.chunk_pairs, opt.cid8, opt.short_seg, opt.row_tile_max, opt.sort_columns = chunk_pairs, cid8, short_seg, row_tile_max, sort_columns
        opt.two_phase, opt.tp_col_block, opt.tp_row_block, opt.long_cb = two_phase, tp_col_block, tp_row_block, long_cb
        self._pb = None
        if part_bounds is not None:
            self._pb = np.ascontiguousarray(part_bounds, np.int32)
            opt.n_parts = self._pb.size - 1
            opt.part_bounds = self._pb.ctypes.data_as(C.POINTER(C.c_int))
            opt.part_stride = int(part_stride)
        self._h = C.c_void_p()
        _lib.check(L.dasp_plan_create(C.byref(self._h), precision, self.rowA, self.colA, self.nnzA, _vp(rp), _vp(ci), _vp(v), C.byref(opt)))
        self.y_order = y_order
        self.x_len = self.colA if part_bounds is None else (self._pb.size - 1) * int(part_stride)

    @classmethod
    def from_device(cls, d_row_ptr, d_col_idx, d_val, rowA, colA, nnzA, precision=64, threshold=0.75, block_longest=256,
                    y_order=Y_PERMUTED, long_piece=0, part_bounds=None, part_stride=0, x_window=0, row_window=0, cid16=0, col_panels=0, slab_max_len=0,
                    x_window_hybrid=0, piece_min_len=0, chunk_pairs=0, cid8=0, short_seg=0, row_tile_max=0, sort_columns=0, two_phase=0, tp_col_block=0, tp_row_block=0, long_cb=0):
        """Plan from a CSR that already lives on the GPU (integer device addresses): packed by kernels, comes back uploaded."""
        L = _lib.lib()
        self = cls.__new__(cls)
        self.precision, self.rowA, self.colA, self.nnzA = precision, int(rowA), int(colA), int(nnzA)
        opt = Options()
        L.dasp_options_default(C.byref(opt))
        opt.threshold, opt.block_longest, opt.y_order, opt.long_piece = threshold, block_longest, y_order, long_piece
        opt.x_window, opt.row_window, opt.cid16, opt.col_panels = x_window, row_window, cid16, col_panels
        opt.slab_max_len, opt.x_window_hybrid, opt.piece_min_len = slab_max_len, x_window_hybrid, piece_min_len
        opt.chunk_pairs, opt.cid8, opt.short_seg, opt.row_tile_max, opt.sort_columns = chunk_pairs, cid8, short_seg, row_tile_max, sort_columns
        opt.two_phase, opt.tp_col_block, opt.tp_row_block, opt.long_cb = two_phase, tp_col_block, tp_row_block, long_cb
        self._pb = None
        if part_bounds is not None:
            self._pb = np.ascontiguousarray(part_bounds, np.int32)
            opt.n_parts, opt.part_stride = self._pb.size - 1, int(part_stride)
            opt.part_bounds = self._pb.ctypes.data_as(C.POINTER(C.c_int))
        self._h = C.c_void_p()
        _lib.check(L.dasp_plan_create_device(C.byref(self._h), precision, self.rowA, self.colA, self.nnzA, C.c_void_p(d_row_ptr),
                                             C.c_void_p(d_col_idx), C.c_void_p(d_val), C.byref(opt)))
        self.y_order = y_order
        self.x_len = int(L.dasp_plan_x_len(self._h))
        return self

    def device_array(self, name, count, dtype):
        """Download one nnz-sized packed array from the device arena (dasp_plan_download_array)."""
        out = np.empty(count, dtype)
        _lib.check(_lib.lib().dasp_plan_download_array(self._h, name.encode(), _vp(out), out.nbytes))
        return out

    def save(self, path):
        """Write the packed plan to `path` (dasp_plan_save)."""
        _lib.check(_lib.lib().dasp_plan_save(self._h, os.fsencode(path)))

    @classmethod
    def load(cls, path):
        """Plan from a file written by save(): ready for upload(), no re-packing."""
        self = cls.__new__(cls)
        self._h = C.c_void_p()
        self._pb = None
        _lib.check(_lib.lib().dasp_plan_load(C.byref(self._h), os.fsencode(path)))
        st = self.stats
        self.precision, self.rowA, self.colA, self.nnzA = st["precision"], st["rowA"], st["colA"], st["nnzA"]
        self.y_order = _lib.lib().dasp_plan_y_order(self._h)
        self.x_len = int(_lib.lib().dasp_plan_x_len(self._h))
        return self

    # -- host side -------------------------------------------------------------------
    @property
    def order_rid(self):
        if self.rowA == 0:
            return np.zeros(0, np.int32)
        p = _lib.lib().dasp_plan_order(self._h)
        return np.ctypeslib.as_array(p, (max(self.rowA, 1),))[: self.rowA].copy()

    @property
    def stats(self):
        s = Stats()
        _lib.check(_lib.lib().dasp_plan_stats(self._h, C.byref(s)))
        return s.as_dict()

    @property
    def n_panels(self):
        return int(_lib.lib().dasp_plan_panel_count(self._h))

    def panel(self, k):
        """Borrowed view of column panel k: (Plan, col_begin, col_end).  Valid while this plan lives."""
        L = _lib.lib()
        h = L.dasp_plan_panel(self._h, k)
        if not h:
            raise _lib.DaspError(-10, L.dasp_last_error().decode("utf-8", "replace"))
        b, e = C.c_int(), C.c_int()
        _lib.check(L.dasp_plan_panel_range(self._h, k, C.byref(b), C.byref(e)))
        sub = Plan.__new__(Plan)
        sub._h, sub._pb, sub._borrowed, sub._parent = C.c_void_p(h), None, True, self
        st = sub.stats
        sub.precision, sub.rowA, sub.colA, sub.nnzA = st["precision"], st["rowA"], st["colA"], st["nnzA"]
        sub.y_order, sub.x_len = Y_NATURAL, st["colA"]
        return sub, b.value, e.value

    def host_array(self, name):
        ptr, eb = C.c_void_p(), C.c_int()
        n = _lib.lib().dasp_plan_host_array(self._h, name.encode(), C.byref(ptr), C.byref(eb))
        if n < 0:
            _lib.check(int(n))
        if name in ("med_cid16", "long_cid16", "rt_start", "tp_lrow", "tp_lcol", "lcb_lcol"):
            dt = np.uint16
        elif name == "rt_mask":
            dt = np.uint64
        elif name == "med_cid8":
            dt = np.uint8
        elif eb.value == 4:
            dt = np.int32
        else:
            dt = np.float64 if eb.value == 8 else np.float16
        if n == 0:
            return np.zeros(0, dt)
        buf = (C.c_char * (n * eb.value)).from_address(ptr.value)
        return np.frombuffer(buf, dtype=dt).copy()

    # -- device side -----------------------------------------------------------------
    def upload(self):
        _lib.check(_lib.lib().dasp_plan_upload(self._h))
        return self

    def tune_placement(self, trials=0, dX=0, dY=0):
        """placement trials (include/dasp_amd.h): (ms of the first allocation, ms of the kept one); (0, 0) if the plan does not qualify.
        dX / dY: the caller's own device vectors as integer addresses (dY is overwritten), 0 = scratch ones.  upload() runs the trials by
        itself (scratch operands) for host-built plans; a plan built from a device CSR calls this when it wants them."""
        a, b = C.c_double(), C.c_double()
        _lib.check(_lib.lib().dasp_plan_tune_placement(self._h, int(trials), C.c_void_p(dX or None), C.c_void_p(dY or None), C.byref(a), C.byref(b)))
        return a.value, b.value

    def drop_host(self):
        _lib.check(_lib.lib().dasp_plan_drop_host(self._h))

    def set_stream_policy(self, policy):
        """0 auto, 1 plain loads (reference dasp_spmv), 2 non-temporal loads (reference dasp_spmv2 'bypass')."""
        _lib.check(_lib.lib().dasp_plan_set_stream_policy(self._h, int(policy)))

    def spmv(self, dX, dY, stream=0, accumulate=False):
        """dX, dY: integer device addresses (e.g. torch_tensor.data_ptr()); stream: hipStream_t as int.
        accumulate: y += A x instead of y = A x (dasp_plan_spmv_acc)."""
        f = _lib.lib().dasp_plan_spmv_acc if accumulate else _lib.lib().dasp_plan_spmv
        _lib.check(f(self._h, C.c_void_p(dX), C.c_void_p(dY), C.c_void_p(stream)))

    def time(self, dX, dY, stream=0, warmup=100, iters=1000):
        """The reference's protocol (dasp_f64.h:1285-1320): returns (wall_ms, event_ms) per SpMV."""
        w, e = C.c_double(), C.c_double()
        _lib.check(_lib.lib().dasp_plan_time(self._h, C.c_void_p(dX), C.c_void_p(dY), C.c_void_p(stream), warmup, iters, C.byref(w), C.byref(e)))
        return w.value, e.value

    def time_each(self, dX, dY, stream=0, warmup=20, iters=200):
        """ms of every one of `iters` back-to-back launches (an event between each two): numpy float32[iters]"""
        out = np.zeros(iters, np.float32)
        _lib.check(_lib.lib().dasp_plan_time_each(self._h, C.c_void_p(dX), C.c_void_p(dY), C.c_void_p(stream), warmup, iters, _vp(out)))
        return out

    def time_graph(self, dX, dY, stream=0, warmup=100, iters=1000, batch=100):
        """Same protocol, `batch` SpMVs captured into one hipGraph and replayed: (wall_ms, event_ms) per SpMV."""
        w, e = C.c_double(), C.c_double()
        _lib.check(_lib.lib().dasp_plan_time_graph(self._h, C.c_void_p(dX), C.c_void_p(dY), C.c_void_p(stream), warmup, iters, batch, C.byref(w), C.byref(e)))
        return w.value, e.value

    def close(self):
        if getattr(self, "_h", None):
            if not getattr(self, "_borrowed", False):
                _lib.lib().dasp_plan_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def spmv_all(filename, csrValA, csrRowPtrA, csrColIdxA, X_val, rowA, colA, nnzA, NUM=4, threshold=0.75, block_longest=256,
             precision=64):
    """One-shot entry with the reference's argument list; returns (Y_val, order_rid):
    Y_val[i] is the product for row order_rid[i].  Runs the reference's 100 + 1000 launches and
    prints its result line."""
    L = _lib.lib()
    dt = _dtype(precision)
    v = np.ascontiguousarray(csrValA, dt)
    rp = np.ascontiguousarray(csrRowPtrA, np.int32)
    ci = np.ascontiguousarray(csrColIdxA, np.int32)
    x = np.ascontiguousarray(X_val, dt)
    y = np.zeros(rowA, dt)
    order = np.zeros(rowA, np.int32)
    fn = L.dasp_spmv_all_f64 if precision == 64 else L.dasp_spmv_all_f16
    _lib.check(fn(os.fsencode(filename or ""), _vp(v), _vp(rp), _vp(ci), _vp(x), _vp(y), _vp(order), rowA, colA, nnzA, NUM, threshold, block_longest))
    return y, order


def partition_rows(csrRowPtr, n_parts):
    """Contiguous row ranges with equal nonzero counts -> int32[n_parts+1]."""
    rp = np.ascontiguousarray(csrRowPtr, np.int32)
    b = np.zeros(n_parts + 1, np.int32)
    _lib.check(_lib.lib().dasp_partition_rows(rp.size - 1, _vp(rp), n_parts, _vp(b)))
    return b


def selftest_mfma():
    _lib.check(_lib.lib().dasp_selftest_mfma())


# ---- synthetic stand-ins for the SuiteSparse inputs (dasp_amd/csrc/gen.cpp) -----------------
SYNTH_NAMES = ("cop20k_A", "nlpkkt160", "powerlaw_1M", "webbase-1M", "ljournal-2008", "HV15R", "Queen_4147", "rmat_2M",
               "webbase-1M-uniform", "ljournal-2008-uniform")


def synth_generator(name):
    """One-line description of the stand-in's generator (seed, structure, locality parameters)."""
    t = _lib.lib().dasp_synth_generator(name.encode())
    if t is None:
        raise _lib.DaspError(-10, _lib.lib().dasp_last_error().decode("utf-8", "replace"))
    return t.decode()


def synth_dims(name, scale=1.0):
    r, c = C.c_int(), C.c_int()
    _lib.check(_lib.lib().dasp_synth_dims(name.encode(), scale, C.byref(r), C.byref(c)))
    return r.value, c.value


def synth_row_lengths(name, scale=1.0, row_begin=0, row_end=None):
    rows, _ = synth_dims(name, scale)
    row_end = rows if row_end is None else row_end
    out = np.zeros(row_end - row_begin, np.int32)
    _lib.check(_lib.lib().dasp_synth_row_lengths(name.encode(), scale, row_begin, row_end, _vp(out)))
    return out


def synth_csr(name, scale=1.0, row_begin=0, row_end=None, lengths=None):
    """CSR pattern (global column ids) of rows [row_begin,row_end): (row_ptr_local, col_idx)."""
    rows, _ = synth_dims(name, scale)
    row_end = rows if row_end is None else row_end
    if lengths is None:
        lengths = synth_row_lengths(name, scale, row_begin, row_end)
    rp = np.zeros(row_end - row_begin + 1, np.int64)
    np.cumsum(lengths, out=rp[1:])
    if rp[-1] >= 2 ** 31:
        raise ValueError("slice has more than 2^31 nonzeros")
    rp = rp.astype(np.int32)
    ci = np.empty(int(rp[-1]), np.int32)
    _lib.check(_lib.lib().dasp_synth_rows(name.encode(), scale, row_begin, row_end, _vp(rp), _vp(ci)))
    return rp, ci

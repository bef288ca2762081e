"""ctypes mirror of the dasp_mg_* part of the C ABI (include/dasp_amd.h): row-partitioned y = A x over several GPUs, one
process per GPU (no reference counterpart: the reference is single-GPU, src/main_f64.cu).

All of the choreography -- the own / other column split, the accumulate launch, the communication stream, the RCCL
all-gather -- lives in dasp_amd/csrc/multigpu.cpp; this file only converts arguments.  Rank r owns the rows
[bounds[r], bounds[r+1]) and, for a square matrix, the x entries of the same indices; after every product the padded y
slices are all-gathered into the buffer the next product reads as x.
"""
import ctypes as C

import numpy as np

from . import _lib
from . import api as D


def _vp(a):
    return a.ctypes.data_as(C.c_void_p)


def unique_id():
    """ncclGetUniqueId: 128 bytes that rank 0 creates and every rank passes to MgPlan.comm_init."""
    buf = np.zeros(128, np.uint8)
    _lib.check(_lib.lib().dasp_mg_unique_id(_vp(buf)))
    return buf


class StreamTimer:
    """HIP events on an arbitrary stream handle (ctypes; torch.cuda.Event only sees torch's current stream, and torch refuses a CU-masked
    stream as its current one): t = StreamTimer(stream); t.start(); ...launches...; ms = t.stop() (synchronises the stop event)."""

    @staticmethod
    def _runtime():
        """the HIP runtime ALREADY mapped into this process (the one libdasp_amd.so and torch use): a second copy found under another
        name would know nothing of their streams"""
        try:
            for line in open("/proc/self/maps"):
                path = line.split()[-1]
                if "libamdhip64.so" in path:
                    return C.CDLL(path)
        except OSError:
            pass
        return C.CDLL("libamdhip64.so")

    def __init__(self, stream):
        _lib.lib()                                   # maps libdasp_amd.so and with it the runtime
        self._hip = self._runtime()
        self._s = C.c_void_p(stream)
        self._e = [C.c_void_p(), C.c_void_p()]
        for e in self._e:
            if self._hip.hipEventCreate(C.byref(e)) != 0:
                raise RuntimeError("hipEventCreate failed")

    def start(self):
        if self._hip.hipEventRecord(self._e[0], self._s) != 0:
            raise RuntimeError("hipEventRecord failed")

    def stop(self):
        ms = C.c_float()
        if self._hip.hipEventRecord(self._e[1], self._s) != 0 or self._hip.hipEventSynchronize(self._e[1]) != 0 or \
                self._hip.hipEventElapsedTime(C.byref(ms), self._e[0], self._e[1]) != 0:
            raise RuntimeError("HIP event timing failed")
        return float(ms.value)

    def __del__(self):
        for e in getattr(self, "_e", []):
            if e:
                self._hip.hipEventDestroy(e)


class MgPlan:
    def __init__(self, rp, ci, val, n_rows, n_cols, bounds, rank, precision=64, overlap=True, threads=0, **opts):
        """rp / ci / val: this rank's CSR slice (local row pointer, GLOBAL column ids)."""
        L = _lib.lib()
        self.precision = precision
        dt = np.float64 if precision == 64 else np.float16
        rp = np.ascontiguousarray(rp, np.int32)
        ci = np.ascontiguousarray(ci, np.int32)
        v = np.ascontiguousarray(val, dt)
        b = np.ascontiguousarray(bounds, np.int32)
        opt = _lib.Options()
        L.dasp_options_default(C.byref(opt))
        opt.host_threads = threads
        for k, x in opts.items():
            setattr(opt, k, x)
        self._h = C.c_void_p()
        _lib.check(L.dasp_mg_plan_create(C.byref(self._h), precision, int(n_rows), int(n_cols), b.size - 1, int(rank), _vp(b), _vp(rp),
                                         _vp(ci), _vp(v), C.byref(opt), int(overlap)))      # 0: one plan, no overlap; 1 / True: own / other column plans; 2: one plan, boundary rows last (the step on one stream)
        self.bounds, self.rank, self.world = b, int(rank), b.size - 1
        i = self.info
        self.stride, self.rows, self.overlap = i["stride"], i["row_end"] - i["row_begin"], bool(i["overlap"])
        self.nnz_local, self.nnz_remote = i["nnz_own"], i["nnz_other"]
        self.n_rows, self.n_cols = int(n_rows), int(n_cols)

    @property
    def info(self):
        s = _lib.MgInfo()
        _lib.check(_lib.lib().dasp_mg_info(self._h, C.byref(s)))
        return s.as_dict()

    def subplan(self, which):
        """Borrowed api.Plan view of the own-column (0) / other-column (1) plan; None if absent."""
        h = _lib.lib().dasp_mg_subplan(self._h, which)
        if not h:
            return None
        sub = D.Plan.__new__(D.Plan)
        sub._h, sub._pb, sub._borrowed, sub._parent = C.c_void_p(h), None, True, self
        st = sub.stats
        sub.precision, sub.rowA, sub.colA, sub.nnzA = st["precision"], st["rowA"], st["colA"], st["nnzA"]
        sub.y_order, sub.x_len = D.Y_NATURAL, int(_lib.lib().dasp_plan_x_len(sub._h))
        return sub

    def upload(self):
        _lib.check(_lib.lib().dasp_mg_upload(self._h))
        return self

    def comm_init(self, uid):
        uid = np.ascontiguousarray(uid, np.uint8)
        assert uid.size == 128
        _lib.check(_lib.lib().dasp_mg_comm_init(self._h, _vp(uid)))

    def set_x(self, x_full):
        dt = np.float64 if self.precision == 64 else np.float16
        x = np.ascontiguousarray(x_full, dt)
        assert x.size == self.n_cols
        _lib.check(_lib.lib().dasp_mg_set_x(self._h, _vp(x)))

    def spmv(self, stream=0):
        _lib.check(_lib.lib().dasp_mg_spmv(self._h, C.c_void_p(stream)))

    def product(self, stream=0):
        _lib.check(_lib.lib().dasp_mg_product(self._h, C.c_void_p(stream)))

    def allgather(self, stream=0):
        """the exchange alone (collective): current y slice -> every rank's gather buffer"""
        _lib.check(_lib.lib().dasp_mg_allgather(self._h, C.c_void_p(stream)))

    def check(self):
        """device sync + did a wait of the fused step time out?  Raises DaspError (and the plan drops to the two-launch form)."""
        _lib.check(_lib.lib().dasp_mg_check(self._h))

    def set_fused(self, on):
        _lib.check(_lib.lib().dasp_mg_set_fused(self._h, 1 if on else 0))

    def set_fake_exchange(self, micros, peers=()):
        """TEST HOOK: n_gpus > 1 without RCCL; the exchange copies this rank's slice into its own gather buffer and into the
        gather buffers of `peers` (MgPlans on the same device), then holds the communication stream for `micros` us."""
        arr = (C.c_void_p * max(1, len(peers)))(*[p if isinstance(p, int) else p.gathered_ptr for p in peers])      # MgPlans, or raw device addresses of gather-sized buffers
        _lib.check(_lib.lib().dasp_mg_set_fake_exchange(self._h, int(micros), len(peers), arr))

    # direct exchange (include/dasp_amd.h): stores into the peers' gather buffers instead of an RCCL collective
    IPC_BYTES = 256

    def push_export(self):
        """this rank's DASP_MG_IPC_BYTES (bytes): all-gather them among the ranks, then push_connect"""
        buf = np.zeros(self.IPC_BYTES, np.uint8)
        _lib.check(_lib.lib().dasp_mg_push_export(self._h, _vp(buf)))
        return buf.tobytes()

    def push_connect(self, blobs):
        """blobs: every rank's push_export in rank order (bytes or a list of bytes); maps the peers and switches the plan over"""
        if not isinstance(blobs, (bytes, bytearray)):
            blobs = b"".join(blobs)
        buf = np.frombuffer(blobs, np.uint8).copy()
        assert buf.size == self.IPC_BYTES * self.world
        _lib.check(_lib.lib().dasp_mg_push_connect(self._h, _vp(buf)))

    def set_exchange(self, mode):
        """'rccl' / 'push' (after push_connect)"""
        _lib.check(_lib.lib().dasp_mg_set_exchange(self._h, {"rccl": 0, "push": 1}[mode]))

    def reserved_stream(self, cus=32):
        """a compute stream of the plan that keeps `cus` CUs free for RCCL's kernels (integer handle, as torch's cuda_stream); None if
        the device cannot make one"""
        h = _lib.lib().dasp_mg_reserved_stream(self._h, int(cus))
        return int(h) if h else None

    def push_loopback(self):
        """TEST HOOK: the direct exchange with scratch memory of this rank standing in for every peer (timing on a one-GPU box)"""
        _lib.check(_lib.lib().dasp_mg_push_loopback(self._h))

    def wait(self, stream=0):
        _lib.check(_lib.lib().dasp_mg_wait(self._h, C.c_void_p(stream)))

    def get_y(self):
        dt = np.float64 if self.precision == 64 else np.float16
        y = np.zeros(self.n_rows, dt)
        _lib.check(_lib.lib().dasp_mg_get_y(self._h, _vp(y)))
        return y

    def get_y_local(self):
        dt = np.float64 if self.precision == 64 else np.float16
        y = np.zeros(self.rows, dt)
        _lib.check(_lib.lib().dasp_mg_get_y_local(self._h, _vp(y)))
        return y

    # integer device addresses (as torch's data_ptr())
    @property
    def y_local_ptr(self):
        return int(_lib.lib().dasp_mg_y_local(self._h) or 0)

    @property
    def gathered_ptr(self):
        return int(_lib.lib().dasp_mg_gathered(self._h) or 0)

    @property
    def x_ptr(self):
        return int(_lib.lib().dasp_mg_x(self._h) or 0)

    def close(self):
        if getattr(self, "_h", None):
            _lib.lib().dasp_mg_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

"""Row-partitioned y = A x over several GPUs, one process per GPU (no reference counterpart: the reference is
single-GPU, src/main_f64.cu).

Rank r owns the rows [bounds[r], bounds[r+1]) and, for a square matrix, the x entries of the same indices.  After every product
the padded y slices are all-gathered into the buffer the next product reads as x.  With `overlap` the rank's nonzeros are split
by column ownership into two DASP plans:

    local  : columns inside the rank's own range -> reads the rank's own slice of x, available as soon as the rank's previous
             product is done;
    remote : everything else -> reads the all-gather buffer (columns remapped into its padded layout by the plan).

so the local product of iteration t+1 runs while the all-gather of iteration t is still in flight; only the (small, for banded
matrices) remote product waits for it.  The exchange is a callback so that the same choreography runs over RCCL
(`all_gather_into_tensor(..., async_op=True)`) and, in tests, through host memory.
"""
import numpy as np

from . import api as D


def split_by_owner(rp, ci, val, lo, hi):
    """CSR slice -> (local, remote): entries with lo <= col < hi (columns re-based to 0) and the rest (global columns)."""
    m = rp.size - 1
    own = (ci >= lo) & (ci < hi)
    rows = np.repeat(np.arange(m, dtype=np.int32), np.diff(rp))

    def sub(mask, shift):
        rp2 = np.zeros(m + 1, np.int64)
        np.cumsum(np.bincount(rows[mask], minlength=m), out=rp2[1:])
        return rp2.astype(np.int32), (ci[mask] - shift).astype(np.int32), val[mask]

    return sub(own, lo), sub(~own, 0)


class RowPartitionedSpMV:
    def __init__(self, torch, rp, ci, val, n_cols, bounds, rank, precision=64, overlap=True, threads=0, stride=None):
        self.torch = torch
        bounds = np.ascontiguousarray(bounds, np.int32)
        self.bounds, self.rank, self.world = bounds, rank, bounds.size - 1
        self.rows = int(bounds[rank + 1] - bounds[rank])
        self.stride = int(stride) if stride else (int(np.diff(bounds).max()) + 63) // 64 * 64
        self.precision = precision
        square = int(bounds[-1]) == int(n_cols)
        self.overlap = bool(overlap) and square
        kw = dict(precision=precision, y_order=D.Y_NATURAL, host_threads=threads)
        self.plan_rem = None
        if self.overlap:
            (rpl, cil, vl), (rpr, cir, vr) = split_by_owner(rp, ci, val, int(bounds[rank]), int(bounds[rank + 1]))
            self.plan = D.Plan(rpl, cil, vl, self.stride, **kw)                   # x = this rank's own padded slice
            if cir.size:
                self.plan_rem = D.Plan(rpr, cir, vr, n_cols, part_bounds=bounds, part_stride=self.stride, **kw)
            self.nnz_local, self.nnz_remote = int(cil.size), int(cir.size)
        else:
            self.plan = D.Plan(rp, ci, val, n_cols, part_bounds=bounds, part_stride=self.stride, **kw)
            self.nnz_local, self.nnz_remote = int(ci.size), 0
        for p in (self.plan, self.plan_rem):
            if p is not None:
                p.upload()
                p.drop_host()
        tdt = torch.float64 if precision == 64 else torch.float16
        z = lambda n: torch.zeros(n, dtype=tdt, device="cuda")
        self.ys = [z(self.stride), z(self.stride)]       # this rank's padded slice of x / y, ping-pong
        self.gathered = z(self.world * self.stride)      # every rank's slice: the x the remote (or whole) plan reads
        self.cur = 0
        self.pending = None

    def seed(self, x_full):
        """x_0 (host array of the n columns / rows) -> this rank's slice and the gathered layout."""
        torch, b, s = self.torch, self.bounds, self.stride
        g = np.zeros(self.world * s, x_full.dtype)
        for r in range(self.world):
            g[r * s: r * s + b[r + 1] - b[r]] = x_full[b[r]:b[r + 1]]
        self.gathered.copy_(torch.from_numpy(g))
        self.ys[0].copy_(self.gathered[self.rank * s:(self.rank + 1) * s])
        self.cur, self.pending = 0, None

    def step(self, gather):
        """One iteration: y = A x, then start the exchange that makes y the next x.  `gather(dst, src)` starts the all-gather of
        the ranks' `src` slices into `dst` and returns an object with .wait() (stream-side) or None if it already completed."""
        torch = self.torch
        s = torch.cuda.current_stream().cuda_stream
        cur, nxt = self.cur, 1 - self.cur
        if self.overlap:
            self.plan.spmv(self.ys[cur].data_ptr(), self.ys[nxt].data_ptr(), s)   # needs only this rank's own x
            if self.pending is not None:
                self.pending.wait()                                               # the other ranks' x has arrived
            if self.plan_rem is not None:                                         # y += (other ranks' columns) * x
                self.plan_rem.spmv(self.gathered.data_ptr(), self.ys[nxt].data_ptr(), s, accumulate=True)
        else:
            if self.pending is not None:
                self.pending.wait()
            self.plan.spmv(self.gathered.data_ptr(), self.ys[nxt].data_ptr(), s)
        self.pending = gather(self.gathered, self.ys[nxt])
        self.cur = nxt

    def finish(self):
        if self.pending is not None:
            self.pending.wait()
            self.pending = None

    @property
    def y_local(self):
        return self.ys[self.cur][: self.rows]

    def full_y(self):
        """The gathered result without the padding (valid after finish())."""
        b, s = self.bounds, self.stride
        return self.torch.cat([self.gathered[r * s: r * s + int(b[r + 1] - b[r])] for r in range(self.world)])

    def close(self):
        for p in (self.plan, self.plan_rem):
            if p is not None:
                p.close()

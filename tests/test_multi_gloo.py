"""N > 1 path on CPU (gloo, world_size 2): row partition by nonzeros, per-rank dasp_mg plans (C ABI, host part) with the
own / other column split and the column remap, all-gather of padded y slices into the layout x is read from.  The local
product is done by decoding the packed plans here (tests may use the oracle; the product path has no CPU fallback) -- what is
under test is the sharding / exchange layout that dasp_mg_spmv drives with RCCL on the GPUs."""
import os
import socket
import sys

import numpy as np
import pytest

import util

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import torch
    import torch.distributed as dist
    import dasp_amd as D
    from oracle import oracle as O
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        m = n = 1200
        rp, ci, v = util.mixed_matrix(m, n, 77)
        bounds = D.partition_rows(rp, world)
        stride = (int(np.diff(bounds).max()) + 63) // 64 * 64
        r0, r1 = int(bounds[rank]), int(bounds[rank + 1])
        sl = slice(rp[r0], rp[r1])
        from dasp_amd.multi import MgPlan
        mg = MgPlan(rp[r0:r1 + 1] - rp[r0], ci[sl], v[sl], m, n, bounds, rank, overlap=True)
        assert mg.stride == stride
        # x in the gathered layout; every rank starts from the same x
        x = np.random.default_rng(5).uniform(-1, 1, n)
        xg = np.zeros(world * stride)
        for g in range(world):
            xg[g * stride: g * stride + bounds[g + 1] - bounds[g]] = x[bounds[g]:bounds[g + 1]]
        x_own = xg[rank * stride:(rank + 1) * stride]
        # own-column plan reads the rank's own slice, other-column plan the gathered layout: y = own + other
        y_local = np.zeros(stride)
        for which, xv in ((0, x_own), (1, xg)):
            sub = mg.subplan(which)
            if sub is None:
                continue
            order = sub.order_rid
            for slot, (cs, vs) in util.decode_plan(sub).items():
                if cs:
                    y_local[order[slot]] += float(np.dot(np.asarray(vs, np.float64), xv[np.asarray(cs, np.int64)]))
        # the one-plan layout of the one-stream step (overlap = 2): ONE plan over all columns in the gathered layout, every row once
        mg2 = MgPlan(rp[r0:r1 + 1] - rp[r0], ci[sl], v[sl], m, n, bounds, rank, overlap=2)
        assert mg2.subplan(1) is None and mg2.nnz_local + mg2.nnz_remote == rp[r1] - rp[r0] and mg2.nnz_local == mg.nnz_local
        sub2 = mg2.subplan(0)
        y2 = np.zeros(stride)
        for slot, (cs, vs) in util.decode_plan(sub2).items():
            if cs:
                y2[sub2.order_rid[slot]] = float(np.dot(np.asarray(vs, np.float64), xg[np.asarray(cs, np.int64)]))
        assert np.abs(y2 - y_local).max() <= 1e-12 * max(np.abs(y_local).max(), 1e-300)
        gathered = torch.zeros(world * stride, dtype=torch.float64)
        dist.all_gather_into_tensor(gathered, torch.from_numpy(y_local))
        full = np.concatenate([gathered.numpy()[g * stride: g * stride + bounds[g + 1] - bounds[g]] for g in range(world)])
        ref = O.csr_spmv(rp, ci, v, x)
        scale = np.maximum(O.csr_absrow(rp, ci, v, x), 1e-300)
        err = float((np.abs(full - ref) / scale).max())
        q.put((rank, err, int(np.diff(rp[bounds]).max()), int(rp[-1])))
    finally:
        dist.destroy_process_group()


def test_two_rank_partition_and_allgather():
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=240) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for rank, err, max_part, nnz in res:
        assert err <= 1e-12
        assert max_part <= nnz / 2 + 2500          # balanced by nonzeros up to one row

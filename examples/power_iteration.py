#!/usr/bin/env python3
"""Power iteration with a DASP plan: the SpMV in its real role (y of one step is x of the next).

  python examples/power_iteration.py [--workload Queen_4147] [--scale 0.02] [--iters 100]
  python -m torch.distributed.run --nproc-per-node N --master-addr 127.0.0.1 examples/power_iteration.py   # row-partitioned

Single GPU: the plan writes y in natural row order (DASP_Y_NATURAL), so y feeds straight back as x.
N GPUs: every rank owns a row range (equal nonzeros); column ids are remapped at pack time into the all-gather layout, so the
buffer `all_gather_into_tensor` fills IS the next x -- no unpacking between iterations.
Values are a seeded function of the pattern (symmetric for the symmetric stand-ins), x0 = 1.
"""
import argparse
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def values_for(rows, ci, rp, seed=7):
    """a_ij = 1 + 0.5 * h(min(i,j), max(i,j)) in [1, 1.5): symmetric whenever the pattern is"""
    r = np.repeat(np.arange(rp.size - 1, dtype=np.int64) + rows, np.diff(rp))
    lo, hi = np.minimum(r, ci), np.maximum(r, ci)
    h = (lo * 1000003 + hi * 7919 + seed) % 104729
    return 1.0 + 0.5 * h / 104729.0


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--workload", default="Queen_4147")
    ap.add_argument("--scale", type=float, default=0.02)
    ap.add_argument("--iters", type=int, default=100)
    args = ap.parse_args()
    import torch
    import dasp_amd as D

    rank, world, local = int(os.environ.get("RANK", 0)), int(os.environ.get("WORLD_SIZE", 1)), int(os.environ.get("LOCAL_RANK", 0))
    torch.cuda.set_device(local)
    dist = None
    if world > 1:
        import torch.distributed as dist
        dist.init_process_group("nccl", device_id=torch.device("cuda", local))
    rows, cols = D.synth_dims(args.workload, args.scale)
    lengths = D.synth_row_lengths(args.workload, args.scale)
    rp_full = np.concatenate([[0], np.cumsum(lengths, dtype=np.int64)])
    if world > 1:
        bounds = np.searchsorted(rp_full, rp_full[-1] * np.arange(world + 1) // world).astype(np.int32)
        bounds[0], bounds[-1] = 0, rows
        stride = (int(np.diff(bounds).max()) + 63) // 64 * 64
        r0, r1 = int(bounds[rank]), int(bounds[rank + 1])
    else:
        bounds, stride, r0, r1 = None, 0, 0, rows
    rp, ci = D.synth_csr(args.workload, args.scale, r0, r1, lengths=lengths[r0:r1])
    val = values_for(r0, ci, rp)
    plan = D.Plan(rp, ci, val, cols, y_order=D.Y_NATURAL, part_bounds=bounds, part_stride=stride).upload()
    x = torch.ones(plan.x_len, dtype=torch.float64, device="cuda")
    y = torch.zeros(max(stride, r1 - r0), dtype=torch.float64, device="cuda")
    nxt = torch.zeros_like(x)
    stream = torch.cuda.current_stream().cuda_stream
    lam = 0.0
    for it in range(args.iters):
        plan.spmv(x.data_ptr(), y.data_ptr(), stream)
        if world > 1:
            dist.all_gather_into_tensor(nxt, y[:stride])          # padded slices: exactly the layout the plan reads
        else:
            nxt[:rows] = y[:rows]
        lam = float(torch.linalg.vector_norm(nxt))               # ||A x|| with ||x|| = 1 (pads are zero)
        x, nxt = nxt / lam, x
    if rank == 0:
        print("%s scale %g, %d rows, %d GPUs: dominant eigenvalue estimate after %d iterations = %.12g" %
              (args.workload, args.scale, rows, world, args.iters, lam))
    if world > 1:
        dist.destroy_process_group()
    return lam


if __name__ == "__main__":
    main()

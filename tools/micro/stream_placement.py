#!/usr/bin/env python3
"""tools/micro/stream_placement.py: does a plain streaming read run at different speeds over different allocations of the same size in one process?
six 2.6-GB buffers, a reduction over each (torch.sum: one pass), interleaved, three rounds; GB/s per buffer."""
import sys, torch
n = 2_649_580_800 // 8
bufs = [torch.ones(n, dtype=torch.float64, device="cuda") for _ in range(6)]
print("addresses", [hex(b.data_ptr()) for b in bufs])
for rnd in range(3):
    line = "round %d:" % rnd
    for b in bufs:
        for _ in range(3): b.sum()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20): b.sum()
        e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 20
        line += "  %.0f" % (n * 8 / ms / 1e6)
    print(line + "  GB/s", flush=True)

#!/usr/bin/env python3
"""Try the plan options that are decided by heuristics (x windows, 16-bit ids, column panels, cache policy) on one matrix and
print what each costs next to the automatic choice -- for real .mtx files, where the heuristics were not tuned.

  tools/autotune.py A.mtx [--precision 64|16]          (or a stand-in name: --workload HV15R --scale 0.3)
"""
import argparse
import itertools
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("mtx", nargs="?")
    ap.add_argument("--workload")
    ap.add_argument("--scale", type=float, default=1.0)
    ap.add_argument("--precision", type=int, default=64, choices=[64, 16])
    ap.add_argument("--iters", type=int, default=100)
    args = ap.parse_args()
    import torch
    import dasp_amd as D
    prec = args.precision
    dt = np.float64 if prec == 64 else np.float16
    if args.mtx:
        m, n, nnz, _, rp, ci, val = D.mmio_allinone(args.mtx, prec)
    else:
        m, n = D.synth_dims(args.workload, args.scale)
        rp, ci = D.synth_csr(args.workload, args.scale)
        val = np.ones(ci.size, dt)
    x = torch.ones(n, dtype=torch.float64 if prec == 64 else torch.float16, device="cuda")
    y = torch.zeros(max(m, 1), dtype=x.dtype, device="cuda")

    def run(**kw):
        plan = D.Plan(rp, ci, val, n, precision=prec, **kw).upload()
        plan.drop_host()
        _, e = plan.time(x.data_ptr(), y.data_ptr(), 0, 10, args.iters)
        st = plan.stats
        plan.close()
        return e, st

    base, st = run()
    print(f"auto: {base*1e3:9.2f} us   windows={st['n_windows_lds']}/{st['n_windows']} cid16={st['cid16_on']} panels={st['n_col_panels']} "
          f"fill0={st['rate_fill0']:.4f} pre={st['pre_ms']:.0f} ms")
    rows = []
    for xw, c16, cp, pol in itertools.product((-1, 81920), (-1, 1), (1, 2, 4), (1, 2)):
        for rt in ((0,) if cp == 1 else (0, -1)):          # column panels: with their row tiles (auto bound) and without
            try:
                e, st = run(x_window=xw, cid16=c16, col_panels=cp, stream_policy=pol, row_tile_max=rt)
            except D.DaspError as exc:
                print("skip", xw, c16, cp, pol, rt, exc)
                continue
            rows.append((e, xw, c16, cp, pol, rt, st))
    rows.sort(key=lambda r: r[0])
    for e, xw, c16, cp, pol, rt, st in rows[:8]:
        print(f"      {e*1e3:9.2f} us   x_window={xw:6d} cid16={c16:2d} col_panels={cp} stream_policy={pol} row_tile_max={rt:2d}   "
              f"(windows {st['n_windows_lds']}/{st['n_windows']}, cid16 {st['cid16_on']}, panels {st['n_col_panels']}, tiles {st['n_row_tiles']})  {e/base:5.2f} x auto")


if __name__ == "__main__":
    main()

"""Worker of test_push_exchange_between_two_processes (tests/test_gpu_spmv.py): rank `rank` of `world` on device 0 of ITS OWN process.
The ranks find each other through files in `dir`: <rank>.blob (dasp_mg_push_export) -> connect -> chained dasp_mg_spmv -> <rank>.npy"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import dasp_amd as D
from dasp_amd.multi import MgPlan

d, rank, world, iters, fused = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4]), int(sys.argv[5])
scale = 0.02
rows, _ = D.synth_dims("HV15R", scale)
rp, ci = D.synth_csr("HV15R", scale)
lens = np.diff(rp)
val = np.repeat(0.5 / np.maximum(lens, 1), lens)
bounds = D.partition_rows(rp, world)
r0, r1 = int(bounds[rank]), int(bounds[rank + 1])
mg = MgPlan(rp[r0:r1 + 1] - rp[r0], ci[rp[r0]:rp[r1]], val[rp[r0]:rp[r1]], rows, rows, bounds, rank, cid16=1).upload()
if not fused:
    mg.set_fused(False)


def publish(name, data):
    with open(os.path.join(d, name + ".tmp"), "wb") as f:
        f.write(data)
    os.rename(os.path.join(d, name + ".tmp"), os.path.join(d, name))


def collect(fmt):
    out, t0 = [], time.time()
    for r in range(world):
        p = os.path.join(d, fmt % r)
        while not os.path.exists(p):
            if time.time() - t0 > 300:
                raise SystemExit("rank %d: no %s after 300 s" % (rank, p))
            time.sleep(0.01)
        out.append(open(p, "rb").read())
    return out


publish("%d.blob" % rank, mg.push_export())
mg.push_connect(collect("%d.blob"))
assert mg.info["exchange"] == 1
x0 = np.random.default_rng(4).uniform(0.5, 1.5, rows)
for rnd, n in enumerate((iters, 3)):          # a second dasp_mg_set_x after an odd / even number of exchanges: both halves of the buffer
    mg.set_x(x0 if rnd == 0 else y)
    for _ in range(n):
        mg.spmv(0)
    mg.check()
    y = mg.get_y()
np.save(os.path.join(d, "%d.out.npy" % rank), y)
publish("%d.done" % rank, b"ok")
collect("%d.done")                            # nobody frees its buffers while a peer may still be storing into them
mg.close()

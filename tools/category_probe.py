import sys, numpy as np, torch
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import dasp_amd as D
name, prec = sys.argv[1], int(sys.argv[2])
m, n = D.synth_dims(name, 1.0)
rp, ci = D.synth_csr(name, 1.0)
lens = np.diff(rp)
dt = np.float64 if prec == 64 else np.float16
x = torch.ones(n, dtype=torch.float64 if prec == 64 else torch.float16, device="cuda")
y = torch.zeros(m, dtype=x.dtype, device="cuda")
def run(tag, keep):
    l2 = np.where(keep, lens, 0)
    rp2 = np.zeros(m + 1, np.int32); np.cumsum(l2, out=rp2[1:])
    sel = np.repeat(keep, lens)
    ci2 = ci[sel]
    plan = D.Plan(rp2, ci2, np.ones(ci2.size, dt), n, precision=prec).upload()
    _, e = plan.time(x.data_ptr(), y.data_ptr(), 0, 200, 2000)
    _, g = plan.time_graph(x.data_ptr(), y.data_ptr(), 0, 200, 2000, 50)
    st = plan.stats
    print(f"{name} {tag}: rows {int(keep.sum())} nnz {ci2.size}  {e*1e3:.2f} us (graph {g*1e3:.2f})  tiles short={st['n_short_tiles']} blocks={st['n_med_blocks']} pieces={st['n_long_pieces']}", flush=True)
    plan.close()
run("all", np.ones(m, bool))
run("short(1-4)", (lens >= 1) & (lens <= 4))
run("len1", lens == 1); run("len2", lens == 2); run("len3", lens == 3); run("len4", lens == 4)
run("medium", (lens >= 5) & (lens < 256))
run("long", lens >= 256)
run("none", np.zeros(m, bool))

import sys, numpy as np, torch
import dasp_amd as D
name, prec = "webbase-1M", 16
m, n = D.synth_dims(name, 1.0)
rp, ci = D.synth_csr(name, 1.0)
lens = np.diff(rp)
dt = np.float16
x = torch.ones(n, dtype=torch.float16, device="cuda")
y = torch.zeros(m, dtype=x.dtype, device="cuda")
keep = (lens >= 5) & (lens < 256)
l2 = np.where(keep, lens, 0)
rp2 = np.zeros(m + 1, np.int32); np.cumsum(l2, out=rp2[1:])
ci2 = ci[np.repeat(keep, lens)]
rows2 = np.repeat(np.arange(m), l2)
def run(tag, cols, **kw):
    plan = D.Plan(rp2, cols.astype(np.int32), np.ones(cols.size, dt), n, precision=prec, **kw).upload()
    _, e = plan.time(x.data_ptr(), y.data_ptr(), 0, 200, 2000)
    st = plan.stats
    print(f"{tag}: {e*1e3:.2f} us  blocks={st['n_med_blocks']} fill0_reg={st['fill0_nnz_reg']} irreg={st['nnz_irreg']} nnz={cols.size}", flush=True)
    plan.close()
run("medium, original columns", ci2)
run("medium, all columns = 0", np.zeros_like(ci2))
run("medium, columns = row id (diagonal-ish)", np.minimum(rows2, n - 1))
run("medium, columns = sequential mod 4096", np.arange(ci2.size) % 4096)
run("medium, original, threshold 0.3", ci2, threshold=0.3)
run("medium, original, threshold 0.01", ci2, threshold=0.01)
run("medium, cols=0, threshold 0.01", np.zeros_like(ci2), threshold=0.01)

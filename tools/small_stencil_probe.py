import numpy as np, scipy.sparse as sp, torch, sys
import dasp_amd as D
for nx in (100, 200, 300, 500, 1000, 2000):
    T = sp.diags([1, 1, 1], [-1, 0, 1], shape=(nx, nx))
    A = (sp.kron(sp.identity(nx), T) + sp.kron(sp.diags([1, 1], [-1, 1], shape=(nx, nx)), sp.identity(nx))).tocsr()
    A.sort_indices()
    rp, ci = A.indptr.astype(np.int32), A.indices.astype(np.int32)
    m = nx * nx
    for prec in (64, 16):
        dt = np.float64 if prec == 64 else np.float16
        x = torch.ones(m, dtype=torch.float64 if prec == 64 else torch.float16, device="cuda")
        y = torch.zeros(m, dtype=x.dtype, device="cuda")
        out = []
        for smax in (4, 0):
            plan = D.Plan(rp, ci, np.ones(ci.size, dt), m, precision=prec, slab_max_len=smax).upload()
            _, e = plan.time(x.data_ptr(), y.data_ptr(), 0, 50, 500)
            st = plan.stats
            out.append(f"{'blocks' if smax == 4 else 'auto  '} {e*1e3:7.2f} us (blocks={st['n_med_blocks']} win={st['n_windows_lds']})")
            plan.close()
        print(f"5-point {nx}^2 f{prec}: " + "   ".join(out), flush=True)

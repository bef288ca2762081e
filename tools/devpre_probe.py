import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.getcwd())
import dasp_amd as D
name = "ljournal-2008"
rows, cols = D.synth_dims(name, 1.0)
rp, ci = D.synth_csr(name, 1.0)
val = np.ones(ci.size, np.float16)
d_rp, d_ci, d_v = torch.from_numpy(rp).cuda(), torch.from_numpy(ci).cuda(), torch.from_numpy(val).cuda()
for T in (-1, 0, -1, 0):
    os.environ["DASP_VERBOSE"] = "1" if T == 0 else ""
    if not os.environ["DASP_VERBOSE"]: del os.environ["DASP_VERBOSE"]
    torch.cuda.synchronize(); t = time.time()
    p = D.Plan.from_device(d_rp.data_ptr(), d_ci.data_ptr(), d_v.data_ptr(), rows, cols, int(rp[-1]), precision=16, row_tile_max=T)
    torch.cuda.synchronize(); print("row_tile_max", T, "device build %.1f ms" % ((time.time() - t) * 1e3), flush=True)
    p.close()

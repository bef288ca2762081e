#!/usr/bin/env python3
"""tools/real_mtx_check.py -- exercises bench.py's DASP_MTX_DIR path: writes a stand-in as a real .mtx, then runs bench.py on it."""
import os, subprocess, sys, tempfile
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import dasp_amd as D
d = tempfile.mkdtemp()
rows, cols = D.synth_dims("Queen_4147", 0.01)
rp, ci = D.synth_csr("Queen_4147", 0.01)
r = np.repeat(np.arange(rows), np.diff(rp))
keep = r >= ci                                    # lower triangle of the symmetric stand-in
with open(os.path.join(d, "Queen_4147.mtx"), "w") as f:
    f.write("%%%%MatrixMarket matrix coordinate pattern symmetric\n%d %d %d\n" % (rows, cols, int(keep.sum())))
    np.savetxt(f, np.column_stack([r[keep] + 1, ci[keep] + 1]), fmt="%d %d")
env = dict(os.environ, DASP_MTX_DIR=d)
out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--workload", "Queen_4147", "--steps", "50", "--no-suite", "--no-cpu-baseline"],
                     env=env, capture_output=True, text=True)
print(out.stdout[-1200:], out.stderr[-500:])
assert '"data": "suitesparse"' in out.stdout and '"verified": true' in out.stdout and str(int(rp[-1])) in out.stdout
print("real-matrix path ok")

#!/usr/bin/env python3
"""tools/block_spread.py [workload=ljournal-2008]: how far apart (in row ids) are the 16 rows of a length-sorted medium block?  The reference's sort is STABLE, so
rows of equal length keep their row order and a block of a common length is 16 nearby rows; only rare lengths are scattered.  CPU only."""
import sys
import numpy as np
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import dasp_amd as D
name = sys.argv[1] if len(sys.argv) > 1 else "ljournal-2008"
lens = D.synth_row_lengths(name, 1.0).astype(np.int64)
med = np.where((lens >= 5) & (lens < 256))[0]
order = med[np.argsort(-lens[med], kind="stable")]
nb = order.size // 16
blk = order[:nb * 16].reshape(nb, 16)
spread = blk.max(1) - blk.min(1)
w = lens[blk].sum(1)
idx = np.argsort(spread); cw = np.cumsum(w[idx]) / w.sum()
for q in (0.1, 0.25, 0.5, 0.75, 0.9):
    print("nnz-weighted quantile %.2f of a block's row-id spread: %d rows" % (q, spread[idx][np.searchsorted(cw, q)]))
print("blocks %d; share of the medium nonzeros in blocks spanning <= 4096 rows: %.3f, <= 65536 rows: %.3f" % (nb, w[spread <= 4096].sum() / w.sum(), w[spread <= 65536].sum() / w.sum()))
for L in (5, 8, 16, 32, 64, 128):
    sel = lens[blk[:, 0]] == L
    if sel.any():
        print("length %3d: %6d blocks, median spread %d rows" % (L, sel.sum(), int(np.median(spread[sel]))))

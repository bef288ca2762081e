#!/usr/bin/env python3
"""tools/id_dict_probe.py [workload ...] -- how many column ids of the medium blocks fit ONE BYTE under three encodings,
on the full-size stand-ins (CPU only, sampled blocks):
  chunk-base : today's cid8 -- all valid ids of a 16 x 4 chunk within 254 of the chunk's minimum
  k-base     : one base per (chunk, k): the 16 rows' k-th columns within 254 of their minimum          (+ 4 ints per chunk = 0.25 B / nnz)
  dict       : per 16-row block a dictionary of <= 255 distinct (col - base_of_row) offsets, base_of_row = the row's first column / the row id
Blocks = 16 consecutive rows of the stable length-descending order of the rows with 5 <= len < 256 (the plan's medium rows)."""
import sys, os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import dasp_amd as D

def probe(name, scale=1.0, every=64):
    rp, ci = D.synth_csr(name, scale)
    rp = rp.astype(np.int64)
    lens = np.diff(rp)
    med = np.nonzero((lens >= 5) & (lens < 256))[0]
    med = med[np.argsort(-lens[med], kind="stable")]
    nb = med.size // 16
    tot = dict(nnz=0, chunk=0, kbase=0, dict_first=0, dict_row=0, blocks=0, dsize_first=[], dsize_row=[])
    for b in range(0, nb, every):
        rows = med[16 * b:16 * b + 16]
        L = int(lens[rows].max())
        nc = (L + 3) // 4
        C = np.full((16, nc * 4), -1, np.int64)
        for i, r in enumerate(rows):
            C[i, :lens[r]] = ci[rp[r]:rp[r + 1]]
        valid = C >= 0
        n = int(valid.sum())
        tot["nnz"] += n; tot["blocks"] += 1
        big = np.where(valid, C, np.iinfo(np.int64).max); small = np.where(valid, C, -1)
        # chunk-base
        Cc = big.reshape(16, nc, 4); Sc = small.reshape(16, nc, 4); Vc = valid.reshape(16, nc, 4)
        span = Sc.max(axis=(0, 2)) - Cc.min(axis=(0, 2))
        tot["chunk"] += int(Vc.sum(axis=(0, 2))[span <= 254].sum())
        # k-base
        spank = small.max(axis=0) - big.min(axis=0)
        tot["kbase"] += int(valid.sum(axis=0)[spank <= 254].sum())
        # k-base, one-byte bases in units of 128 columns above the chunk's minimum: per-k span <= 127 and every k-minimum < 255 * 128 above the chunk's
        kmin = big.min(axis=0).reshape(nc, 4); kspan = spank.reshape(nc, 4); kval = valid.sum(axis=0).reshape(nc, 4)
        has = kval > 0
        cmin = np.where(has, kmin, np.iinfo(np.int64).max).min(axis=1, keepdims=True)
        ok = (~has | ((kspan <= 127) & (kmin - cmin < 255 * 128))).all(axis=1)
        tot["kbase128"] = tot.get("kbase128", 0) + int(kval.sum(axis=1)[ok].sum())
        ok16 = (~has | ((kspan <= 254) & (kmin - cmin < 65535))).all(axis=1)
        tot["kbase16"] = tot.get("kbase16", 0) + int(kval.sum(axis=1)[ok16].sum())
        # dictionaries
        first = C[:, 0:1]
        for key, base in (("first", first), ("row", rows.reshape(16, 1))):
            offs = np.unique((C - base)[valid])
            tot["dsize_" + key].append(offs.size)
            if offs.size <= 255: tot["dict_" + key] += n
    n = max(tot["nnz"], 1)
    print("%-20s k-base by whole chunks: u16 bases, span 254: %.3f ; u8 bases x128, span 127: %.3f" % (name, tot["kbase16"] / n, tot["kbase128"] / n))
    print("%-20s blocks sampled %6d nnz %9d | one-byte share: chunk-base %.3f  k-base %.3f  dict(first col) %.3f  dict(row id) %.3f | dict size median first %d row %d, p90 %d / %d"
          % (name, tot["blocks"], n, tot["chunk"] / n, tot["kbase"] / n, tot["dict_first"] / n, tot["dict_row"] / n,
             np.median(tot["dsize_first"]), np.median(tot["dsize_row"]), np.percentile(tot["dsize_first"], 90), np.percentile(tot["dsize_row"], 90)))

if __name__ == "__main__":
    for w in (sys.argv[1:] or ["nlpkkt160", "HV15R", "Queen_4147", "HV15R-unstructured", "cop20k_A"]):
        probe(w)

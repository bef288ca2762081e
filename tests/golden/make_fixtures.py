"""Writes the tiny MatrixMarket fixtures and expected.json next to this script.

The reference's loader cannot be compiled in this image (src/common.h needs cuSPARSE/cuBLAS
headers), so the expected CSR / classifier values are produced by the CPU oracle
(oracle/dasp_oracle.c): they pin the PRODUCT against the ORACLE and guard the oracle against
regressions; they are not outputs of the reference ("parity unpinned", see oracle/dasp_oracle.h).
Two entries ARE reference-derived: `radix_known_answer` (recorded by the survey from a run of the
real utils.h radix_sort) and the banner/size codes, which tests re-check against
oracle/_ref/libref_mmio.so (the reference's own mmio.h).
Run:  python tests/golden/make_fixtures.py
"""
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
from oracle import oracle as O  # noqa: E402

FILES = {
    "gen_real.mtx": "%%MatrixMarket matrix coordinate real general\n% comment\n4 5 6\n1 1 1.5\n2 3 -2.25\n4 5 1e-3\n2 1 4\n4 1 7.0\n2 3 8.5\n",
    "sym_real.mtx": "%%MatrixMarket matrix coordinate real symmetric\n4 4 5\n1 1 1.0\n3 1 2.0\n2 2 3.0\n4 2 4.0\n4 4 5.0\n",
    "pattern_sym.mtx": "%%MatrixMarket MATRIX Coordinate Pattern SYMMETRIC\n%a\n%b\n5 5 4\n2 1\n3 3\n5 1\n5 4\n",
    "integer_gen.mtx": "%%MatrixMarket matrix coordinate integer general\n3 3 4\n1 2 -7\n3 3 12\n1 1 3\n2 2 0\n",
    "complex_herm.mtx": "%%MatrixMarket matrix coordinate complex hermitian\n3 3 3\n1 1 2.0 0.0\n2 1 1.5 -0.5\n3 3 4.0 0\n",
    "skew.mtx": "%%MatrixMarket matrix coordinate real skew-symmetric\n3 3 2\n2 1 1.0\n3 2 -2.0\n",
    "blank_before_size.mtx": "%%MatrixMarket matrix coordinate real general\n%c\n\n\n2 2 2\n1 1 1\n2 2 2\n",
    "multi_per_line.mtx": "%%MatrixMarket matrix coordinate real general\n3 3 3\n1 1 1.0 2 2\n2.0\n3 3 3.0\n",
    "empty_rows.mtx": "%%MatrixMarket matrix coordinate pattern general\n6 4 3\n2 1\n2 4\n5 2\n",
    "crlf.mtx": "%%MatrixMarket matrix coordinate real general\r\n2 2 2\r\n1 2 0.5\r\n2 1 0.25\r\n",
    "bad_banner.mtx": "%MatrixMarket matrix coordinate real general\n1 1 0\n",
    "bad_field.mtx": "%%MatrixMarket matrix coordinate double general\n1 1 0\n",
    "no_size.mtx": "%%MatrixMarket matrix coordinate real general\n% only comments\n",
}


def main():
    exp = {}
    for name, text in FILES.items():
        p = os.path.join(HERE, name)
        with open(p, "w", newline="") as f:
            f.write(text)
        rc_b, tc = O.mm_read_banner(p)
        rc_s, M, N, nz = O.mm_read_size(p)
        rc, m, n, nnz, sym, rp, ci, v = O.mmio_allinone(p)
        e = {"banner_rc": rc_b, "typecode": tc, "size_rc": rc_s, "size": [M, N, nz], "rc": rc}
        if rc == 0:
            e.update(m=m, n=n, nnz=nnz, sym=sym, row_ptr=rp.tolist(), col_idx=ci.tolist(), val=v.tolist())
        exp[name] = e
    exp["radix_known_answer"] = {"key": [5, 17, 5, 200, 9, 17], "idx": [0, 1, 2, 3, 4, 5],
                                 "key_sorted": [200, 17, 17, 9, 5, 5], "idx_sorted": [3, 1, 5, 4, 0, 2]}
    exp["sym4_survey"] = {"file": "sym_real.mtx", "row_ptr": [0, 2, 4, 5, 7], "col_idx": [0, 2, 1, 3, 0, 1, 3]}
    # classifier / packer regression values on seeded matrices (tests/util.py builders)
    sys.path.insert(0, os.path.dirname(HERE))
    import util
    cases = {}
    for tag, (builder, m, n, seed) in {"mixed": (util.mixed_matrix, 3000, 2500, 7), "pairs": (util.pair_heavy_matrix, 4000, 3000, 11)}.items():
        rp, ci, v = builder(m, n, seed)
        for prec in (64, 16):
            P = O.Packed(prec, rp, ci, v, n)
            fields = "row_long row_block row_zero rowloop short_row_1 short_row_2 short_row_3 short_row_4 common_13 nnz_short nnz_long origin_nnz_reg nnz_irreg fill0_nnz_short fill0_nnz_long fill0_nnz_reg blocknum warp_number".split()
            cases["%s_f%d" % (tag, prec)] = dict({f: int(getattr(P, f)) for f in fields}, order_fnv="%016x" % O.fnv1a_i32(P.order_rid),
                                                  builder=tag, m=m, n=n, seed=seed)
    exp["packer_cases"] = cases
    with open(os.path.join(HERE, "expected.json"), "w") as f:
        json.dump(exp, f, indent=1, sort_keys=True)
    print("wrote", len(FILES), "fixtures + expected.json")


if __name__ == "__main__":
    main()

"""Writes tests/golden/handworked.json: two small matrices whose classifier counters and order_rid were worked out BY HAND
from the reference's source lines (no oracle, no product code involved) so that the pin is auditable on paper.

Reference lines followed (paths relative to the reference tree):
  classifier test order        src/dasp_f64.h:499-531 (len 1, 3, 2, 0, 4, >= block_longest, else medium)   f16: 1029-1061
  rowloop                      src/dasp_f64.h:533-536
  medium sort                  radix_sort(rptA, ridA): descending by length, stable      src/utils.h:118-160,196-203
  1&3 pairing count            src/dasp_f64.h:597-607 : c = min(n1, n3); if c / 8 >= 16: c = 8 * (c / 8), n1 -= c, n3 -= c, else c = 0
                               src/dasp_f16.h:1127-1137: same test (c / 8 >= 16) but c = 32 * (c / 32)
  order_rid, f64               src/dasp_f64.h:960-976 : long | medium(sorted) | short_rid_1[0..n1) |
                                 for i < c/8: 8 x short_rid_1[n1 + 8i + j] then 8 x short_rid_3[8i + j] |
                                 short_rid_3[c..c+n3) | short_rid_4 | short_rid_2 | zero_rid
  order_rid, f16               src/dasp_f16.h:1253-1270: long | medium(sorted) |
                                 for i < c/32: 32 x short_rid_1[n1 + 32i + j] then 32 x short_rid_3[32i + j] |
                                 short_rid_3[c..c+n3) | short_rid_4 | short_rid_2 | short_rid_1[0..n1) | zero_rid
  (n1, n3 above are the counts AFTER the pairing subtraction, as in the reference's variables short_row_1 / short_row_3.)

Case "small24" (24 rows, block_longest 256), row: length
   0:3  1:1  2:0  3:7  4:2  5:256  6:4  7:12  8:1  9:7  10:300  11:3  12:2  13:0  14:5  15:12  16:4  17:1  18:255  19:3  20:7
   21:2  22:9  23:4
  len 1 -> rows 1 8 17 ; len 3 -> 0 11 19 ; len 2 -> 4 12 21 ; len 0 -> 2 13 ; len 4 -> 6 16 23 ; long (>= 256) -> 5 10 ;
  medium (row order) 3(7) 7(12) 9(7) 14(5) 15(12) 18(255) 20(7) 22(9) ; sorted desc, stable: 18 | 7 15 | 22 | 3 9 20 | 14.
  c = min(3,3) = 3, 3/8 = 0 < 16 -> c = 0 in both precisions.  nnz_short = 3*1 + 3*3 + 3*2 + 3*4 = 30, nnz_long = 556,
  rowloop = 1 (8 medium rows < 59990).
  f64: 5 10 | 18 7 15 22 3 9 20 14 | 1 8 17 | 0 11 19 | 6 16 23 | 4 12 21 | 2 13
  f16: 5 10 | 18 7 15 22 3 9 20 14 | 0 11 19 | 6 16 23 | 4 12 21 | 1 8 17 | 2 13

  Regular / irregular split of the one block of 8 sorted medium rows, lengths 255 12 12 9 7 7 7 5 (src/dasp_f64.h:1044-1091: chunk k of a block is
  kept while sum over its rows of min(4, max(0, len - 4(k-1))) >= 0.75 * 32 = 24):
    k=1: 4 x 8 = 32 -> kept ; k=2: 4+4+4+4+3+3+3+1 = 26 -> kept ; k=3: 4+4+4+1+0+0+0+0 = 13 -> stop.
    blockPtr[0] = 2 * 32 = 64 ; irreg_len = max(len - 8, 0) = 247 4 4 1 0 0 0 0 -> irreg_rpt (scan) 0 247 251 255 256 256 256 256 256, nnz_irreg = 256 ;
    blocknum = ceil(8/8) = 1 rounded up to a multiple of 4 * rowloop = 4 (:1045) -> blockPtr (scan) 0 64 64 64 64, fill0_nnz_reg = 64 (f64).
    f16 rounds every block's span up to a multiple of 128 (src/dasp_f16.h:1356): blockPtr 0 128 128 128 128, fill0_nnz_reg = 128.
    origin_nnz_reg = nnzA - nnz_irreg - nnz_long - nnz_short = 900 - 256 - 556 - 30 = 58 (:1091)  (= 8+8+8+8+7+7+7+5).
  Long rows 256 and 300 (src/dasp_f64.h:1000-1039): ceil(len / 64) = 4 and 5 warps -> long_rpt_new 0 4 9, 9 warps rounded up to a multiple of 4 = 12,
    fill0_nnz_long = 12 * 64 = 768 (f64).  f16: groups of 256 (src/dasp_f16.h:1273-1314): 1 and 2 warps -> 0 1 3, rounded to 4, fill0_nnz_long = 1024.

Case "pairs300" (300 rows): rows 0..271: length 1 if the row id is even, 3 if odd (136 of each: short_rid_1 = 0 2 4 .. 270,
  short_rid_3 = 1 3 5 .. 271); 272..279: length 2; 280..287: length 4; 288..291: length 0; 292..295: length 6; 296: length 300;
  297..299: length 9.
  c = min(136, 136) = 136; 136 / 8 = 17 >= 16.
    f64: c = 8 * 17 = 136, n1 = n3 = 0.    f16: c = 32 * (136 / 32) = 128, n1 = n3 = 8.
  medium sorted: 297 298 299 (9) then 292 293 294 295 (6).  nnz_short = 136 + 3*136 + 2*8 + 4*8 = 592, nnz_long = 300.
  f64: 296 | 297 298 299 292 293 294 295 | (no unpaired len-1) | for i in 0..16: [2(8i+j) for j<8] + [2(8i+j)+1 for j<8] |
       (no unpaired len-3) | 280..287 | 272..279 | 288..291
  f16: 296 | 297 298 299 292 293 294 295 | for i in 0..3: [2(8+32i+j) for j<32] + [2(32i+j)+1 for j<32] |
       short_rid_3[128..135] = 257 259 .. 271 | 280..287 | 272..279 | short_rid_1[0..7] = 0 2 .. 14 | 288..291
"""
import json
import os

small_len = [3, 1, 0, 7, 2, 256, 4, 12, 1, 7, 300, 3, 2, 0, 5, 12, 4, 1, 255, 3, 7, 2, 9, 4]
small = {
    "block_longest": 256, "lengths": small_len,
    "counters": {"row_long": 2, "row_block": 8, "row_zero": 2, "short_row_1": 3, "short_row_3": 3, "short_row_2": 3, "short_row_4": 3,
                 "common_13": 0, "nnz_short": 30, "nnz_long": 556, "rowloop": 1},
    "packer_f64": {"nnz_irreg": 256, "origin_nnz_reg": 58, "fill0_nnz_reg": 64, "blocknum": 4, "fill0_nnz_long": 768, "warp_number": 12,
                   "block_ptr": [0, 64, 64, 64, 64], "irreg_rpt": [0, 247, 251, 255, 256, 256, 256, 256, 256], "long_rpt_new": [0, 4, 9]},
    "packer_f16": {"nnz_irreg": 256, "origin_nnz_reg": 58, "fill0_nnz_reg": 128, "blocknum": 4, "fill0_nnz_long": 1024, "warp_number": 4,
                   "block_ptr": [0, 128, 128, 128, 128], "irreg_rpt": [0, 247, 251, 255, 256, 256, 256, 256, 256], "long_rpt_new": [0, 1, 3]},
    "order_f64": [5, 10, 18, 7, 15, 22, 3, 9, 20, 14, 1, 8, 17, 0, 11, 19, 6, 16, 23, 4, 12, 21, 2, 13],
    "order_f16": [5, 10, 18, 7, 15, 22, 3, 9, 20, 14, 0, 11, 19, 6, 16, 23, 4, 12, 21, 1, 8, 17, 2, 13],
}

pl = [1 if i % 2 == 0 else 3 for i in range(272)] + [2] * 8 + [4] * 8 + [0] * 4 + [6] * 4 + [300] + [9] * 3
med = [297, 298, 299, 292, 293, 294, 295]
o64 = [296] + med
for i in range(17):
    o64 += [2 * (8 * i + j) for j in range(8)] + [2 * (8 * i + j) + 1 for j in range(8)]
o64 += list(range(280, 288)) + list(range(272, 280)) + list(range(288, 292))
o16 = [296] + med
for i in range(4):
    o16 += [2 * (8 + 32 * i + j) for j in range(32)] + [2 * (32 * i + j) + 1 for j in range(32)]
o16 += list(range(257, 272, 2)) + list(range(280, 288)) + list(range(272, 280)) + list(range(0, 16, 2)) + list(range(288, 292))
assert sorted(o64) == list(range(300)) and sorted(o16) == list(range(300))
pairs = {
    "block_longest": 256, "lengths": pl,
    "counters_f64": {"row_long": 1, "row_block": 7, "row_zero": 4, "short_row_1": 0, "short_row_3": 0, "short_row_2": 8, "short_row_4": 8,
                     "common_13": 136, "nnz_short": 592, "nnz_long": 300, "rowloop": 1},
    "counters_f16": {"row_long": 1, "row_block": 7, "row_zero": 4, "short_row_1": 8, "short_row_3": 8, "short_row_2": 8, "short_row_4": 8,
                     "common_13": 128, "nnz_short": 592, "nnz_long": 300, "rowloop": 1},
    "order_f64": o64, "order_f16": o16,
}
out = {"_comment": "hand-derived from the reference source, see make_handworked.py's docstring for the derivation", "small24": small, "pairs300": pairs}
json.dump(out, open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "handworked.json"), "w"), indent=0)

// dasp_amd_shim.h -- put next to main_f64.cu / main_f16.cu in place of "dasp_f64.h" / "dasp_f16.h"; link with -ldasp_amd.
// Compile main_f64 with -Df64 (any C++ compiler), main_f16 with -Df16 (a compiler with a half type: hipcc / clang++).
// The two free functions keep the reference's signatures (src/mmio_highlevel.h:608-610, src/dasp_f64.h:486-487,
// src/dasp_f16.h:1015-1016), so no call site of main_f64.cu / main_f16.cu changes: half values cross the C ABI as IEEE binary16
// bit patterns (uint16_t), and the cast lives HERE, once, not at the call sites.
#include <stdint.h>
#include <stdio.h>
#include "dasp_amd.h"
#define MAT_PTR_TYPE int
#if defined(f16)
typedef _Float16 half;                   /* what <cuda_fp16.h> gives the reference: a 2-byte IEEE binary16 arithmetic type */
#define MAT_VAL_TYPE half
#define DASP_SHIM_LOAD(m, n, nnz, s, rp, ci, v, f) dasp_mmio_allinone_f16(m, n, nnz, s, rp, ci, reinterpret_cast<uint16_t **>(v), f)
#define DASP_SHIM_SPMV(f, v, rp, ci, x, y, o, r, c, z, num, th, bl) \
    dasp_spmv_all_f16(f, reinterpret_cast<const uint16_t *>(v), rp, ci, reinterpret_cast<const uint16_t *>(x), reinterpret_cast<uint16_t *>(y), o, r, c, z, num, th, bl)
#else
#define MAT_VAL_TYPE double
#define DASP_SHIM_LOAD(m, n, nnz, s, rp, ci, v, f) dasp_mmio_allinone_f64(m, n, nnz, s, rp, ci, v, f)
#define DASP_SHIM_SPMV dasp_spmv_all_f64
#endif

static inline int mmio_allinone(int *m, int *n, MAT_PTR_TYPE *nnz, int *isSymmetric,
                                MAT_PTR_TYPE **csrRowPtr, int **csrColIdx, MAT_VAL_TYPE **csrVal, char *filename)
{   /* same out-params, same return codes (0 / -1 / -2 / -4); arrays are malloc'd, free() works */
    return DASP_SHIM_LOAD(m, n, nnz, isSymmetric, csrRowPtr, csrColIdx, csrVal, filename);
}

static inline void spmv_all(char *filename, MAT_VAL_TYPE *csrValA, MAT_PTR_TYPE *csrRowPtrA, int *csrColIdxA,
                            MAT_VAL_TYPE *X_val, MAT_VAL_TYPE *Y_val, int *order_rid,
                            int rowA, int colA, MAT_PTR_TYPE nnzA, int NUM, double threshold, int block_longest)
{   /* runs the reference's 100 warm-up + 1000 timed launches, prints its "SpMV_X:" line, appends its CSV row */
    int rc = DASP_SHIM_SPMV(filename, csrValA, csrRowPtrA, csrColIdxA, X_val, Y_val, order_rid,
                            rowA, colA, nnzA, NUM, threshold, block_longest);
    if (rc) fprintf(stderr, "dasp_spmv_all: %d (%s)\n", rc, dasp_last_error());
}
static inline void initVec(MAT_VAL_TYPE *vec, int length) { for (int i = 0; i < length; ++i) vec[i] = 1; } /* utils.h:93-100 */

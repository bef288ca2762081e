// main_shim.cpp -- a driver in the shape of the reference's main() (src/main_f64.cu:102-168, src/main_f16.cu:102-164), written
// against the reference's OWN two entry points and nothing else, to show that dasp_amd_shim.h is a drop-in: load the .mtx, set
// values and x to one (initVec), spmv_all, then compare Y_val with a host CSR product THROUGH order_rid, the way the reference's
// verify_new does (main_f64.cu:3-16: |y_ref[order_rid[i]] - Y_val[i]| against 1e-5; main_f16.cu:5-18: against 1.0).
//   g++     -Df64 main_shim.cpp -I<repo>/include -L<repo>/dasp_amd -ldasp_amd -o main_f64
//   clang++ -Df16 main_shim.cpp -I<repo>/include -L<repo>/dasp_amd -ldasp_amd -o main_f16      (needs a half type: hipcc / clang++)
#include <stdlib.h>
#include <string.h>
#include "dasp_amd_shim.h"

static int verify_through_order(const MAT_VAL_TYPE *y_ref, const MAT_VAL_TYPE *Y_val, const int *order_rid, int length, double tol)
{
    int bad = 0;
    for (int i = 0; i < length; ++i) {
        const double d = (double)y_ref[order_rid[i]] - (double)Y_val[i];
        if (d > tol || d < -tol) ++bad;
    }
    return bad;
}

int main(int argc, char **argv)
{
    if (argc < 2) { printf("Run the code by './main_shim matrix.mtx'.\n"); return 0; }
    char *filename = argv[1];
    int rowA, colA, isSymmetricA;
    MAT_PTR_TYPE nnzA;
    MAT_PTR_TYPE *csrRowPtrA;
    int *csrColIdxA;
    MAT_VAL_TYPE *csrValA;
    const int rc = mmio_allinone(&rowA, &colA, &nnzA, &isSymmetricA, &csrRowPtrA, &csrColIdxA, &csrValA, filename);
    if (rc != 0) { fprintf(stderr, "mmio_allinone: %d\n", rc); return 2; }
    MAT_VAL_TYPE *X_val = (MAT_VAL_TYPE *)malloc(sizeof(MAT_VAL_TYPE) * (size_t)(colA + 1));     /* +1: the reference's f16 kernel over-reads X by one */
    initVec(X_val, colA);
    initVec(csrValA, nnzA);
    MAT_VAL_TYPE *Y_val = (MAT_VAL_TYPE *)malloc(sizeof(MAT_VAL_TYPE) * (size_t)(rowA > 0 ? rowA : 1));
    int *order_rid = (int *)malloc(sizeof(int) * (size_t)(rowA > 0 ? rowA : 1));
    const int NUM = 4, block_longest = 256;
    const double threshold = 0.75;
    spmv_all(filename, csrValA, csrRowPtrA, csrColIdxA, X_val, Y_val, order_rid, rowA, colA, nnzA, NUM, threshold, block_longest);

    /* comparator: serial CSR product on the host (the reference compares with cuSPARSE, main_f64.cu:18-100) */
    MAT_VAL_TYPE *y_ref = (MAT_VAL_TYPE *)malloc(sizeof(MAT_VAL_TYPE) * (size_t)(rowA > 0 ? rowA : 1));
    for (int i = 0; i < rowA; ++i) {
        double s = 0;
        for (int j = csrRowPtrA[i]; j < csrRowPtrA[i + 1]; ++j) s += (double)csrValA[j] * (double)X_val[csrColIdxA[j]];
        y_ref[i] = (MAT_VAL_TYPE)s;
    }
#if defined(f16)
    const double tol = 1.0;
#else
    const double tol = 1e-5;
#endif
    const int bad = verify_through_order(y_ref, Y_val, order_rid, rowA, tol);
    printf("main_shim: rows %d cols %d nnz %d | %s (%d rows beyond %g)\n", rowA, colA, nnzA, bad ? "check FAILED" : "check passed", bad, tol);
    free(X_val); free(Y_val); free(order_rid); free(y_ref); free(csrRowPtrA); free(csrColIdxA); free(csrValA);
    return bad ? 1 : 0;
}

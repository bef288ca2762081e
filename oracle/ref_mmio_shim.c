/*
 * oracle/ref_mmio_shim.c -- TEST INFRASTRUCTURE ONLY.
 * Compiles the reference's OWN src/mmio.h (NIST Matrix Market I/O, self-contained:
 * only libc headers) from where it lies under /root/reference, into
 * oracle/_ref/libref_mmio.so.  Nothing is copied: the include path is given by
 * oracle/Makefile (-I$(REF)/src).  The rest of the reference's hot path cannot be built
 * here (src/common.h:15-16 needs <cusparse.h>/<cublas_v2.h>, absent from this image).
 *
 * Exports (from mmio.h): mm_read_banner, mm_read_mtx_crd_size, mm_read_mtx_crd_data,
 * mm_read_mtx_crd_entry, plus the helpers below that open a path so that ctypes callers
 * need no FILE*.
 */
#include "mmio.h"

int ref_mm_read_banner_path(const char *path, char typecode[4])
{
    FILE *f = fopen(path, "r");
    if (!f) return MM_COULD_NOT_READ_FILE;
    MM_typecode tc;
    int rc = mm_read_banner(f, &tc);
    typecode[0] = tc[0]; typecode[1] = tc[1]; typecode[2] = tc[2]; typecode[3] = tc[3];
    fclose(f);
    return rc;
}

int ref_mm_read_size_path(const char *path, int *M, int *N, int *nz)
{
    FILE *f = fopen(path, "r");
    if (!f) return MM_COULD_NOT_READ_FILE;
    MM_typecode tc;
    int rc = mm_read_banner(f, &tc);
    if (rc == 0) rc = mm_read_mtx_crd_size(f, M, N, nz);
    fclose(f);
    return rc;
}

/* the reference's bulk entry parser (src/mmio.h:866-923): I, J 1-based as in the file; val holds nz reals, or 2*nz
 * (re, im) pairs for complex; pattern leaves val untouched; integer matrices are MM_UNSUPPORTED_TYPE there. */
int ref_mm_read_crd_data_path(const char *path, int cap, int *I, int *J, double *val, int *nz_out)
{
    FILE *f = fopen(path, "r");
    if (!f) return MM_COULD_NOT_READ_FILE;
    MM_typecode tc;
    int M, N, nz;
    int rc = mm_read_banner(f, &tc);
    if (rc == 0) rc = mm_read_mtx_crd_size(f, &M, &N, &nz);
    if (rc == 0) {
        *nz_out = nz;
        if (nz > cap) rc = -100;
        else rc = mm_read_mtx_crd_data(f, M, N, nz, I, J, val, tc);
    }
    fclose(f);
    return rc;
}

/* the same entries through the reference's one-at-a-time parser (src/mmio.h:925-980) */
int ref_mm_read_crd_entries_path(const char *path, int cap, int *I, int *J, double *re, double *im, int *nz_out)
{
    FILE *f = fopen(path, "r");
    if (!f) return MM_COULD_NOT_READ_FILE;
    MM_typecode tc;
    int M, N, nz;
    int rc = mm_read_banner(f, &tc);
    if (rc == 0) rc = mm_read_mtx_crd_size(f, &M, &N, &nz);
    if (rc == 0) {
        *nz_out = nz;
        if (nz > cap) rc = -100;
        for (int e = 0; rc == 0 && e < nz; ++e) {
            re[e] = 0.0; im[e] = 0.0;
            rc = mm_read_mtx_crd_entry(f, &I[e], &J[e], &re[e], &im[e], tc);
        }
    }
    fclose(f);
    return rc;
}

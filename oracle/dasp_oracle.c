/*
 * oracle/dasp_oracle.c -- TEST INFRASTRUCTURE ONLY (see dasp_oracle.h for the
 * pinning status: "parity unpinned" except the mmio.h parse layer, radix_sort and the survey-recorded
 * classifier / padded sizes / order_rid hash of one 3000-row matrix).
 *
 * Plain-C restatement of the reference's host algorithm for the DASP SpMV path.
 * Every function cites the reference lines it follows (paths relative to
 * /root/reference/).  Geometry here is the REFERENCE's (8-row blocks, 8x4 MMA tiles,
 * 32-lane warps); the product under dasp_amd/ uses its own CDNA4 geometry and is
 * compared with this file only through geometry-independent outputs (CSR, category
 * counts, order_rid, y).
 */
#include "dasp_oracle.h"

#include <ctype.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

/* reference constants: src/common.h:28-33, src/dasp_f64.h:18-22, src/dasp_f16.h:16-20 */
enum { BS = 8, MK = 4, MM = 8, TILE = MM * MK, WARPS = 4, LOOP_LONG = 2, LOOP_SHORT = 4, GROUPNUM = 1 };

void oracle_free(void *p) { free(p); }

/* ------------------------------------------------------------------ loader */

static void lower_inplace(char *s) { for (; *s; ++s) *s = (char)tolower((unsigned char)*s); }

/* src/mmio.h:398-564 (mm_read_banner): first line, five tokens, case-folded fields 2-5 */
static int banner_from_file(FILE *f, char tc[4])
{
    char line[1025], banner[1025], mtx[1025], crd[1025], dtype[1025], scheme[1025];
    tc[0] = tc[1] = tc[2] = ' '; tc[3] = 'G';
    if (!fgets(line, sizeof line, f)) return 12;
    if (sscanf(line, "%s %s %s %s %s", banner, mtx, crd, dtype, scheme) != 5) return 12;
    lower_inplace(mtx); lower_inplace(crd); lower_inplace(dtype); lower_inplace(scheme);
    if (strncmp(banner, "%%MatrixMarket", 14) != 0) return 14;
    if (strcmp(mtx, "matrix") != 0) return 15;
    tc[0] = 'M';
    if (strcmp(crd, "coordinate") == 0) tc[1] = 'C';
    else if (strcmp(crd, "array") == 0) tc[1] = 'A';
    else return 15;
    if (strcmp(dtype, "real") == 0) tc[2] = 'R';
    else if (strcmp(dtype, "complex") == 0) tc[2] = 'C';
    else if (strcmp(dtype, "pattern") == 0) tc[2] = 'P';
    else if (strcmp(dtype, "integer") == 0) tc[2] = 'I';
    else return 15;
    if (strcmp(scheme, "general") == 0) tc[3] = 'G';
    else if (strcmp(scheme, "symmetric") == 0) tc[3] = 'S';
    else if (strcmp(scheme, "hermitian") == 0) tc[3] = 'H';
    else if (strcmp(scheme, "skew-symmetric") == 0) tc[3] = 'K';
    else return 15;
    return 0;
}

/* src/mmio.h:568-624 (mm_read_mtx_crd_size): skip '%' lines, then "M N nz" */
static int size_from_file(FILE *f, int *M, int *N, int *nz)
{
    char line[1025];
    *M = *N = *nz = 0;
    do {
        if (!fgets(line, sizeof line, f)) return 12;
    } while (line[0] == '%');
    if (sscanf(line, "%d %d %d", M, N, nz) == 3) return 0;
    for (;;) {
        int got = fscanf(f, "%d %d %d", M, N, nz);
        if (got == EOF) return 12;
        if (got == 3) return 0;
    }
}

int oracle_mm_read_banner(const char *path, char typecode[4])
{
    FILE *f = fopen(path, "r");
    if (!f) return 11;
    int rc = banner_from_file(f, typecode);
    fclose(f);
    return rc;
}

int oracle_mm_read_size(const char *path, int *M, int *N, int *nz)
{
    char tc[4];
    FILE *f = fopen(path, "r");
    if (!f) return 11;
    int rc = banner_from_file(f, tc);
    if (rc == 0) rc = size_from_file(f, M, N, nz);
    fclose(f);
    return rc;
}

/* src/mmio_highlevel.h:10-25 */
void oracle_exclusive_scan(int *a, int len)
{
    if (len == 0 || len == 1) return;
    int carry = 0;
    for (int i = 0; i < len; ++i) { int v = a[i]; a[i] = carry; carry += v; }
}

/* the entry loop of mmio_allinone, src/mmio_highlevel.h:663-697, on its own: file-order COO triples (0-based, as after the
 * loop's --i / --j), value = real part / integer / 1.0 for pattern; `im` (may be NULL) receives the imaginary part that
 * mmio_allinone drops.  Pinned against the reference's own entry parsers mm_read_mtx_crd_data / mm_read_mtx_crd_entry
 * (src/mmio.h:866-980) in tests/test_oracle.py.  Arrays are malloc'd (oracle_free). */
static int coo_from_file(FILE *f, const char tc[4], int nz_file, int *ri, int *ci, double *vv, double *vi)
{
    const int is_pattern = tc[2] == 'P', is_real = tc[2] == 'R', is_complex = tc[2] == 'C',
              is_integer = tc[2] == 'I';                /* :632-635 */
    for (int e = 0; e < nz_file; ++e) {                 /* :663-697 */
        int i = 0, j = 0, iv = 0;
        double re = 0.0, im = 0.0;
        if (is_real) { if (fscanf(f, "%d %d %lg\n", &i, &j, &re) < 0) {} }
        else if (is_complex) { if (fscanf(f, "%d %d %lg %lg\n", &i, &j, &re, &im) < 0) {} }
        else if (is_integer) { if (fscanf(f, "%d %d %d\n", &i, &j, &iv) < 0) {} re = iv; }
        else if (is_pattern) { if (fscanf(f, "%d %d\n", &i, &j) < 0) {} re = 1.0; }
        --i; --j;
        ri[e] = i; ci[e] = j; vv[e] = re;
        if (vi) vi[e] = im;
    }
    return 0;
}

int oracle_mm_read_coo(const char *path, char typecode[4], int *M, int *N, int *nz, int **I, int **J, double **re, double **im)
{
    FILE *f = fopen(path, "r");
    if (!f) return -1;
    if (banner_from_file(f, typecode) != 0) { fclose(f); return -2; }
    if (size_from_file(f, M, N, nz) != 0) { fclose(f); return -4; }
    const size_t n = (size_t)(*nz > 0 ? *nz : 1);
    *I = (int *)malloc(sizeof(int) * n); *J = (int *)malloc(sizeof(int) * n);
    *re = (double *)malloc(sizeof(double) * n); *im = (double *)malloc(sizeof(double) * n);
    coo_from_file(f, typecode, *nz, *I, *J, *re, *im);
    fclose(f);
    return 0;
}

/* src/mmio_highlevel.h:608-774 (mmio_allinone) */
int oracle_mmio_allinone(const char *path, int *m, int *n, int *nnz, int *is_symmetric,
                         int **row_ptr, int **col_idx, double **val)
{
    char tc[4];
    int M, N, nz_file;
    FILE *f = fopen(path, "r");
    if (!f) return -1;                                  /* :623-624 */
    if (banner_from_file(f, tc) != 0) {                 /* :626-630 (file is left open there too) */
        printf("Could not process Matrix Market banner.\n");
        fclose(f);
        return -2;
    }
    if (size_from_file(f, &M, &N, &nz_file) != 0) { fclose(f); return -4; } /* :638-640 */
    const int sym = (tc[3] == 'S' || tc[3] == 'H');     /* :642: skew is NOT mirrored */

    int *cnt = (int *)calloc((size_t)M + 1, sizeof(int));
    int *ri = (int *)malloc(sizeof(int) * (size_t)(nz_file > 0 ? nz_file : 1));
    int *ci = (int *)malloc(sizeof(int) * (size_t)(nz_file > 0 ? nz_file : 1));
    double *vv = (double *)malloc(sizeof(double) * (size_t)(nz_file > 0 ? nz_file : 1));

    coo_from_file(f, tc, nz_file, ri, ci, vv, NULL);    /* :663-697 */
    for (int e = 0; e < nz_file; ++e) cnt[ri[e]]++;
    fclose(f);

    if (sym)                                            /* :702-709 */
        for (int e = 0; e < nz_file; ++e)
            if (ri[e] != ci[e]) cnt[ci[e]]++;

    oracle_exclusive_scan(cnt, M + 1);                  /* :712 */
    const int total = cnt[M];
    int *rp = (int *)malloc(sizeof(int) * ((size_t)M + 1));
    memcpy(rp, cnt, sizeof(int) * ((size_t)M + 1));
    int *cid = (int *)malloc(sizeof(int) * (size_t)(total > 0 ? total : 1));
    double *v = (double *)malloc(sizeof(double) * (size_t)(total > 0 ? total : 1));
    memset(cnt, 0, sizeof(int) * ((size_t)M + 1));

    for (int e = 0; e < nz_file; ++e) {                 /* :722-756: file order, mirror right after */
        int r = ri[e], c = ci[e];
        int at = rp[r] + cnt[r]++;
        cid[at] = c; v[at] = vv[e];
        if (sym && r != c) {
            at = rp[c] + cnt[c]++;
            cid[at] = r; v[at] = vv[e];
        }
    }
    free(cnt); free(ri); free(ci); free(vv);
    *m = M; *n = N; *nnz = total; *is_symmetric = sym;
    *row_ptr = rp; *col_idx = cid; *val = v;
    return 0;
}

/* --------------------------------------------------------------- utilities */

/* src/utils.h:118-160 (get_max, count_sort) and :196-203 (radix_sort):
 * LSD base-10 passes; each pass places keys so that larger digits come first and
 * equal digits keep their order => descending, stable. */
void oracle_radix_sort_desc(int *key, int *idx, int len)
{
    if (len <= 0) return;           /* the reference reads arr[0] unguarded; callers pass len>0 or skip */
    int mx = key[0];
    for (int i = 1; i < len; ++i) if (key[i] > mx) mx = key[i];
    int *tk = (int *)malloc(sizeof(int) * (size_t)len), *ti = (int *)malloc(sizeof(int) * (size_t)len);
    for (int e = 1; mx / e > 0; e *= 10) {
        int b[10] = {0};
        for (int i = 0; i < len; ++i) b[(key[i] / e) % 10]++;
        for (int d = 1; d < 10; ++d) b[d] += b[d - 1];
        for (int i = 0; i < len; ++i) {
            int d = (key[i] / e) % 10;
            int pos = len - (b[d] - 1) - 1;
            tk[pos] = key[i]; ti[pos] = idx[i];
            b[d]--;
        }
        memcpy(key, tk, sizeof(int) * (size_t)len);
        memcpy(idx, ti, sizeof(int) * (size_t)len);
        if (e > 214748364) break;   /* int overflow guard; never reached for len<256 keys */
    }
    free(tk); free(ti);
}

void oracle_init_vec_f64(double *v, int len) { for (int i = 0; i < len; ++i) v[i] = 1; }

void oracle_csr_spmv_f64(int m, const int *rp, const int *ci, const double *val, const double *x, double *y)
{
    for (int i = 0; i < m; ++i) {
        double s = 0.0;
        for (int j = rp[i]; j < rp[i + 1]; ++j) s += val[j] * x[ci[j]];
        y[i] = s;
    }
}

void oracle_csr_absrow_f64(int m, const int *rp, const int *ci, const double *val, const double *x, double *sabs)
{
    for (int i = 0; i < m; ++i) {
        double s = 0.0;
        for (int j = rp[i]; j < rp[i + 1]; ++j) s += fabs(val[j] * x[ci[j]]);
        sabs[i] = s;
    }
}

double oracle_round_f16(double v)
{
    if (isnan(v) || isinf(v)) return v;
    double a = fabs(v);
    if (a == 0.0) return v;
    int e;
    frexp(a, &e);                       /* a = f * 2^e, f in [0.5,1) */
    int q = e - 11;                     /* keep 11 significant bits */
    if (q < -24) q = -24;               /* subnormal spacing 2^-24 */
    double scaled = ldexp(a, -q);
    double r = nearbyint(scaled);       /* default rounding mode: to nearest even */
    double out = ldexp(r, q);
    if (out > 65504.0) out = INFINITY;
    return v < 0 ? -out : out;
}

unsigned long long oracle_fnv1a_i32(const int *a, long long len)
{
    unsigned long long h = 1469598103934665603ULL;
    const unsigned char *p = (const unsigned char *)a;
    for (long long i = 0; i < len * 4; ++i) { h ^= p[i]; h *= 1099511628211ULL; }
    return h;
}

/* ------------------------------------------------------- DASP preprocessing */

static int ceil_div(int a, int b) { return (a + b - 1) / b; }

oracle_dasp_t *oracle_dasp_new(void) { return (oracle_dasp_t *)calloc(1, sizeof(oracle_dasp_t)); }

void oracle_dasp_free(oracle_dasp_t *d)
{
    if (!d) return;
    free(d->order_rid); free(d->short_val); free(d->short_cid); free(d->long_val); free(d->long_cid);
    free(d->long_rpt_new); free(d->reg_val); free(d->reg_cid); free(d->block_ptr);
    free(d->irreg_val); free(d->irreg_cid); free(d->irreg_rpt);
    memset(d, 0, sizeof *d);
}

static void *zalloc(size_t n, size_t sz) { return calloc(n ? n : 1, sz); }

int oracle_dasp_pack(int precision, int rowA, int colA, int nnzA,
                     const int *rp, const int *ci, const double *val,
                     double threshold, int block_longest, oracle_dasp_t *o)
{
    if (!o || (precision != 64 && precision != 16) || rowA < 0) return -1;
    memset(o, 0, sizeof *o);
    const int f16 = precision == 16;
    const int sv = f16 ? 2 : 8;
    o->precision = precision; o->rowA = rowA; o->colA = colA; o->nnzA = nnzA;

    /* pass 1: counts.  dasp_f64.h:499-531 / dasp_f16.h:1029-1061 (this test order) */
    int n1 = 0, n2 = 0, n3 = 0, n4 = 0, nz0 = 0, nlong = 0, nmed = 0;
    for (int i = 0; i < rowA; ++i) {
        int len = rp[i + 1] - rp[i];
        if (len == 1) n1++; else if (len == 3) n3++; else if (len == 2) n2++;
        else if (len == 0) nz0++; else if (len == 4) n4++;
        else if (len >= block_longest) nlong++; else nmed++;
    }
    o->rowloop = nmed < 59990 ? 1 : (nmed < 400000 ? 2 : 4);  /* dasp_f64.h:533-536 */

    int *rid1 = (int *)zalloc(n1, 4), *rid2 = (int *)zalloc(n2, 4), *rid3 = (int *)zalloc(n3, 4),
        *rid4 = (int *)zalloc(n4, 4), *ridL = (int *)zalloc(nlong, 4), *rid0 = (int *)zalloc(nz0, 4),
        *ridM = (int *)zalloc(nmed, 4);
    int *rptM = (int *)zalloc((size_t)nmed + 1, 4), *rptL = (int *)zalloc((size_t)nlong + 1, 4);
    {   /* pass 2: row-id lists in row order.  dasp_f64.h:552-594 */
        int a = 0, b = 0, c = 0, d = 0, e = 0, z = 0, g = 0;
        for (int i = 0; i < rowA; ++i) {
            int len = rp[i + 1] - rp[i];
            if (len == 1) rid1[a++] = i; else if (len == 3) rid3[c++] = i; else if (len == 2) rid2[b++] = i;
            else if (len == 0) rid0[z++] = i; else if (len == 4) rid4[d++] = i;
            else if (len >= block_longest) { rptL[e] = len; ridL[e++] = i; }
            else { rptM[g] = len; ridM[g++] = i; }
        }
    }
    const int nnz_short = n1 + 3 * n3 + 2 * n2 + 4 * n4;     /* dasp_f64.h:595 */

    /* 1&3 pairing.  dasp_f64.h:597-607 ; dasp_f16.h:1127-1137 rounds to 4*BlockSize */
    int c13 = n1 < n3 ? n1 : n3;
    if (c13 / BS >= 16) {
        c13 = f16 ? BS * 4 * (c13 / (BS * 4)) : BS * (c13 / BS);
        n1 -= c13; n3 -= c13;
    } else c13 = 0;

    /* tile counts and padded sizes.  dasp_f64.h:609-634 ; dasp_f16.h:1139-1160 */
    const int sb13 = ceil_div(c13, BS);
    const int sb22 = ceil_div((n2 + 1) / 2, BS);
    const int n34 = n3 + n4;
    const int sb34 = ceil_div(n34, BS);
    const int per13 = WARPS * GROUPNUM * (f16 ? 4 : 2), per22 = per13, per34 = WARPS * GROUPNUM * LOOP_SHORT;
    const int tb13 = ceil_div(sb13, per13), tb22 = ceil_div(sb22, per22), tb34 = ceil_div(sb34, per34);
    const int f13 = tb13 * per13 * TILE, f34 = tb34 * per34 * TILE, f22 = tb22 * per22 * TILE;
    const int seg1 = f16 ? ((n1 + 1) / 2) * 2 : n1;
    const int fshort = seg1 + f13 + f34 + f22;
    /* segment offsets inside short_*: f64 [1 | 13 | 34 | 22], f16 [13 | 34 | 22 | 1] */
    const int off1 = f16 ? f13 + f34 + f22 : 0;
    const int off13 = f16 ? 0 : n1;
    const int off34 = off13 + f13, off22 = off34 + f34;

    double *sval = (double *)zalloc(fshort, 8);
    int *scid = (int *)zalloc(fshort, 4);

    for (int i = 0; i < n1; ++i) {                      /* dasp_f64.h:639-644 ; dasp_f16.h:1234-1241 */
        int r = rid1[i];
        sval[off1 + i] = val[rp[r]]; scid[off1 + i] = ci[rp[r]];
    }
    for (int t = 0; t < c13; ++t) {                     /* dasp_f64.h:646-664: tile row t = [v1 | v3 v3 v3] */
        int r1 = rid1[n1 + t], r3 = rid3[t];
        int e0 = off13 + t * MK;
        sval[e0] = val[rp[r1]]; scid[e0] = ci[rp[r1]];
        for (int k = 0; k < 3; ++k) { sval[e0 + 1 + k] = val[rp[r3] + k]; scid[e0 + 1 + k] = ci[rp[r3] + k]; }
    }
    for (int i = 0; i < n3; ++i) {                      /* dasp_f64.h:666-680 */
        int r = rid3[c13 + i], e0 = off34 + i * MK;
        for (int k = 0; k < 3; ++k) { sval[e0 + k] = val[rp[r] + k]; scid[e0 + k] = ci[rp[r] + k]; }
    }
    for (int i = 0; i < n4; ++i) {                      /* dasp_f64.h:682-698 */
        int r = rid4[i], e0 = off34 + (n3 + i) * MK;
        for (int k = 0; k < 4; ++k) { sval[e0 + k] = val[rp[r] + k]; scid[e0 + k] = ci[rp[r] + k]; }
    }
    {   /* 2&2: dasp_f64.h:700-713 groups of 16 rows over one 8-row tile;
         * dasp_f16.h:1217-1232 groups of 64 rows over four tiles */
        const int rows_half = f16 ? BS * 4 : BS;        /* tile rows per group */
        for (int j = 0; j < n2; ++j) {
            int g = j / (2 * rows_half), jj = j % (2 * rows_half);
            int e0 = off22 + g * rows_half * MK + (jj % rows_half) * MK + (jj / rows_half) * 2;
            int r = rid2[j];
            sval[e0] = val[rp[r]]; sval[e0 + 1] = val[rp[r] + 1];
            scid[e0] = ci[rp[r]]; scid[e0 + 1] = ci[rp[r] + 1];
        }
    }

    /* sort medium rows by length, descending & stable.  dasp_f64.h:914 */
    if (nmed > 0) oracle_radix_sort_desc(rptM, ridM, nmed);
    oracle_exclusive_scan(rptM, nmed + 1);              /* :954 */
    oracle_exclusive_scan(rptL, nlong + 1);             /* :955 */
    const int nnz_long = nlong > 0 ? rptL[nlong] : 0;
    /* exclusive_scan leaves len<=1 arrays untouched: nlong==0 => rptL[0]==0 already;
     * nmed==0 => rptM[0]==0 already. */

    /* order_rid.  dasp_f64.h:960-976 ; dasp_f16.h:1253-1270 */
    int *ord = (int *)zalloc(rowA, 4);
    {
        int p = 0;
        memcpy(ord + p, ridL, sizeof(int) * (size_t)nlong); p += nlong;
        memcpy(ord + p, ridM, sizeof(int) * (size_t)nmed); p += nmed;
        if (!f16) { memcpy(ord + p, rid1, sizeof(int) * (size_t)n1); p += n1; }
        const int grp = f16 ? BS * 4 : BS;
        /* f64 iterates short_block13 tiles (c13 is a multiple of 8, so all are full);
         * f16 iterates c13/32 groups */
        for (int g = 0; g < c13 / grp; ++g)
            for (int j = 0; j < grp; ++j) {
                ord[p + g * 2 * grp + j] = rid1[n1 + g * grp + j];
                ord[p + g * 2 * grp + grp + j] = rid3[g * grp + j];
            }
        p += 2 * c13;
        memcpy(ord + p, rid3 + c13, sizeof(int) * (size_t)n3); p += n3;
        memcpy(ord + p, rid4, sizeof(int) * (size_t)n4); p += n4;
        memcpy(ord + p, rid2, sizeof(int) * (size_t)n2); p += n2;
        if (f16) { memcpy(ord + p, rid1, sizeof(int) * (size_t)n1); p += n1; }
        memcpy(ord + p, rid0, sizeof(int) * (size_t)nz0); p += nz0;
    }

    /* long rows.  dasp_f64.h:1000-1039 ; dasp_f16.h:1273-1314 (one warp = 64 / 256 elements) */
    const int G = TILE * LOOP_LONG * (f16 ? 4 : 1);
    int *lrn = (int *)zalloc((size_t)nlong + 1, 4);
    for (int i = 0; i < nlong; ++i) lrn[i] = ceil_div(rptL[i + 1] - rptL[i], G);
    oracle_exclusive_scan(lrn, nlong + 1);
    int warp_number = nlong > 0 ? lrn[nlong] : 0;
    const int bn_long = ceil_div(warp_number, WARPS);
    const int flong = bn_long * WARPS * G;
    warp_number = bn_long * WARPS;
    double *lval = (double *)zalloc(flong, 8);
    int *lcid = (int *)zalloc(flong, 4);
    for (int i = 0; i < nlong; ++i) {
        int r = ridL[i], len = rptL[i + 1] - rptL[i];
        for (int j = 0; j < len; ++j) { lval[(size_t)lrn[i] * G + j] = val[rp[r] + j]; lcid[(size_t)lrn[i] * G + j] = ci[rp[r] + j]; }
    }

    /* regular / irregular split.  dasp_f64.h:1044-1091 ; dasp_f16.h:1317-1365 */
    int blocknum = ceil_div(nmed, BS);
    blocknum = ceil_div(blocknum, o->rowloop * 4) * o->rowloop * 4;
    int *bptr = (int *)zalloc((size_t)blocknum + 1, 4);
    int *irpt = (int *)zalloc((size_t)nmed + 1, 4);
    for (int b = 0; b < blocknum; ++b) {
        int r0 = b * BS, r1 = (b + 1) * BS >= nmed ? nmed : (b + 1) * BS;
        for (int k = 1;; ++k) {
            int fill = 0;
            for (int r = r0; r < r1; ++r) {
                int len = rptM[r + 1] - rptM[r];
                if (len / MK >= k) fill += MK;
                else if (len / MK == k - 1) fill += len % MK;
            }
            if (fill >= threshold * MK * MM) bptr[b] += TILE;
            else {
                for (int r = r0; r < r1; ++r) {
                    int len = rptM[r + 1] - rptM[r];
                    int rest = len - (k - 1) * MK;
                    irpt[r] = rest > 0 ? rest : 0;
                }
                break;
            }
        }
        if (f16) bptr[b] = ceil_div(bptr[b], TILE * 4) * TILE * 4;   /* dasp_f16.h:1356 */
    }
    oracle_exclusive_scan(bptr, blocknum + 1);
    oracle_exclusive_scan(irpt, nmed + 1);
    const int freg = blocknum > 0 ? bptr[blocknum] : 0;
    const int nirr = nmed > 0 ? irpt[nmed] : 0;

    /* irregular tails = LAST irreg_len entries of the row.  dasp_f64.h:1094-1106 */
    double *ival = (double *)zalloc(nirr, 8);
    int *icid = (int *)zalloc(nirr, 4);
    for (int r = 0; r < nmed; ++r) {
        int row = ridM[r], len = irpt[r + 1] - irpt[r];
        for (int j = 0; j < len; ++j) {
            ival[irpt[r] + j] = val[rp[row + 1] - len + j];
            icid[irpt[r] + j] = ci[rp[row + 1] - len + j];
        }
    }

    /* regular tiles, chunk-major [chunk][row][k].  dasp_f64.h:1109-1157 ; dasp_f16.h:1385-1443 */
    double *rval = (double *)zalloc(freg, 8);
    int *rcid = (int *)zalloc(freg, 4);
    for (int b = 0; b < blocknum; ++b) {
        int span = bptr[b + 1] - bptr[b], blen = span / BS;
        for (int rr = 0; rr < BS; ++rr) {
            int r = b * BS + rr;
            int row = r < nmed ? ridM[r] : -1;
            int len = 0;
            if (row >= 0) {
                len = rp[row + 1] - rp[row];
                if (f16) len -= irpt[r + 1] - irpt[r];          /* dasp_f16.h:1402 */
            }
            for (int i = 0; i < blen; ++i) {
                /* flat index rr*blen+i is re-laid by dasp_f64.h:1149 */
                int at = bptr[b] + (i / MK) * BS * MK + rr * MK + i % MK;
                if (row >= 0 && i < len) { rval[at] = val[rp[row] + i]; rcid[at] = ci[rp[row] + i]; }
                else { rval[at] = 0.0; rcid[at] = 0; }
            }
        }
    }

    /* accounting.  dasp_f64.h:1089-1091,1159-1166 ; dasp_f16.h:1448-1455 (irregular values padded to even) */
    o->row_long = nlong; o->row_block = nmed; o->row_zero = nz0;
    o->short_row_1 = n1; o->short_row_2 = n2; o->short_row_3 = n3; o->short_row_4 = n4;
    o->common_13 = c13; o->short_row_34 = n34;
    o->nnz_short = nnz_short; o->nnz_long = nnz_long; o->nnz_irreg = nirr;
    o->origin_nnz_reg = nnzA - nirr - nnz_long - nnz_short;
    o->fill0_nnz_short13 = f13; o->fill0_nnz_short34 = f34; o->fill0_nnz_short22 = f22;
    o->fill0_nnz_short = fshort; o->fill0_nnz_long = flong; o->fill0_nnz_reg = freg;
    o->threadblock13 = tb13; o->threadblock34 = tb34; o->threadblock22 = tb22;
    o->blocknum = blocknum; o->warp_number = warp_number; o->BlockNum_long = bn_long;
    o->offset_short1 = off1;
    {
        long long fill0 = (long long)fshort + flong + nirr + freg;
        o->rate_fill0 = nnzA > 0 ? (double)(fill0 - nnzA) / nnzA : 0.0;
        long long nirr_b = f16 ? ((nirr + 1) / 2) * 2 : nirr;
        o->data_X = (long long)(rowA + colA) * sv + (long long)flong * (sv + 4) + (long long)warp_number * sv +
                    (long long)(nlong + 1) * 4 + (long long)fshort * (sv + 4) + (long long)freg * (sv + 4) +
                    (long long)(blocknum + 1) * 4 + nirr_b * (sv + 4) + (long long)(nmed + 1) * 4;
    }
    o->order_rid = ord;
    o->short_val = sval; o->short_cid = scid;
    o->long_val = lval; o->long_cid = lcid; o->long_rpt_new = lrn;
    o->reg_val = rval; o->reg_cid = rcid; o->block_ptr = bptr;
    o->irreg_val = ival; o->irreg_cid = icid; o->irreg_rpt = irpt;

    free(rid1); free(rid2); free(rid3); free(rid4); free(ridL); free(rid0); free(ridM);
    free(rptM); free(rptL);
    return 0;
}

/* What the fused kernel + longPart_sum leave in y, per category, evaluated from the packed
 * arrays (dasp_f64.h:53-75 long stage 2, :90-144 long, :145-279 medium, :281-295 len-1,
 * :296-356 1&3, :357-423 3&4, :424-483 2&2; f16 counterparts dasp_f16.h:106-590). */
void oracle_dasp_eval(const oracle_dasp_t *d, const double *x, double *y)
{
    const int f16 = d->precision == 16;
    const int G = TILE * LOOP_LONG * (f16 ? 4 : 1);
    for (int i = 0; i < d->rowA; ++i) y[i] = 0.0;       /* cudaMemset(dY_val): dasp_f64.h:1242 */

    /* long: per-warp partials, then one sum per row */
    for (int i = 0; i < d->row_long; ++i) {
        double row = 0.0;
        for (int w = d->long_rpt_new[i]; w < d->long_rpt_new[i + 1]; ++w) {
            double part = 0.0;
            for (int e = 0; e < G; ++e) part += d->long_val[(size_t)w * G + e] * x[d->long_cid[(size_t)w * G + e]];
            row += part;
        }
        y[i] = row;
    }
    /* medium: regular tile rows (diagonal of the MMA) + irregular tail */
    const int base_m = d->row_long;
    for (int b = 0; b < d->blocknum; ++b) {
        for (int p = d->block_ptr[b]; p < d->block_ptr[b + 1]; ++p) {
            int q = p - d->block_ptr[b];
            int rr = (q % (BS * MK)) / MK;
            int r = b * BS + rr;
            if (r < d->row_block) y[base_m + r] += d->reg_val[p] * x[d->reg_cid[p]];
        }
    }
    for (int r = 0; r < d->row_block; ++r)
        for (int p = d->irreg_rpt[r]; p < d->irreg_rpt[r + 1]; ++p)
            y[base_m + r] += d->irreg_val[p] * x[d->irreg_cid[p]];

    /* short segments */
    const int n1 = d->short_row_1, c13 = d->common_13, n34 = d->short_row_34, n2 = d->short_row_2;
    const int base_s = d->row_long + d->row_block;
    const int off1 = d->offset_short1;
    const int off13 = f16 ? 0 : n1;
    const int off34 = off13 + d->fill0_nnz_short13, off22 = off34 + d->fill0_nnz_short34;
    const int y1 = f16 ? base_s + 2 * c13 + n34 + n2 : base_s;
    const int y13 = f16 ? base_s : base_s + n1;
    const int y34 = y13 + 2 * c13, y22 = y34 + n34;
    for (int t = 0; t < n1; ++t) y[y1 + t] = d->short_val[off1 + t] * x[d->short_cid[off1 + t]];
    const int grp = f16 ? BS * 4 : BS;
    for (int t = 0; t < c13; ++t) {
        int e0 = off13 + t * MK;
        int s = y13 + (t / grp) * 2 * grp + t % grp;
        y[s] = d->short_val[e0] * x[d->short_cid[e0]];
        double a = 0.0;
        for (int k = 1; k < 4; ++k) a += d->short_val[e0 + k] * x[d->short_cid[e0 + k]];
        y[s + grp] = a;
    }
    for (int i = 0; i < n34; ++i) {
        double a = 0.0;
        for (int k = 0; k < 4; ++k) a += d->short_val[off34 + i * MK + k] * x[d->short_cid[off34 + i * MK + k]];
        y[y34 + i] = a;
    }
    for (int j = 0; j < n2; ++j) {
        int g = j / (2 * grp), jj = j % (2 * grp);
        int e0 = off22 + g * grp * MK + (jj % grp) * MK + (jj / grp) * 2;
        y[y22 + j] = d->short_val[e0] * x[d->short_cid[e0]] + d->short_val[e0 + 1] * x[d->short_cid[e0 + 1]];
    }
}

/* ------------------------------------------------------------- accessors */

#define FIELD_INT(n) if (strcmp(name, #n) == 0) return d->n
int oracle_dasp_int(const oracle_dasp_t *d, const char *name)
{
    FIELD_INT(precision); FIELD_INT(rowA); FIELD_INT(colA); FIELD_INT(nnzA);
    FIELD_INT(row_long); FIELD_INT(row_block); FIELD_INT(row_zero); FIELD_INT(rowloop);
    FIELD_INT(short_row_1); FIELD_INT(short_row_2); FIELD_INT(short_row_3); FIELD_INT(short_row_4);
    FIELD_INT(common_13); FIELD_INT(short_row_34);
    FIELD_INT(nnz_short); FIELD_INT(nnz_long); FIELD_INT(origin_nnz_reg); FIELD_INT(nnz_irreg);
    FIELD_INT(fill0_nnz_short13); FIELD_INT(fill0_nnz_short34); FIELD_INT(fill0_nnz_short22);
    FIELD_INT(fill0_nnz_short); FIELD_INT(fill0_nnz_long); FIELD_INT(fill0_nnz_reg);
    FIELD_INT(threadblock13); FIELD_INT(threadblock34); FIELD_INT(threadblock22);
    FIELD_INT(blocknum); FIELD_INT(warp_number); FIELD_INT(BlockNum_long); FIELD_INT(offset_short1);
    return -2147483647;
}

/* data_X (dasp_f64.h:1162-1166) and rate_fill0 (:1159-1160) do not fit oracle_dasp_int */
long long oracle_dasp_data_X(const oracle_dasp_t *d) { return d->data_X; }
double oracle_dasp_rate_fill0(const oracle_dasp_t *d) { return d->rate_fill0; }

#define FIELD_ARR(n, l) if (strcmp(name, #n) == 0) { if (len) *len = (l); return d->n; }
const void *oracle_dasp_arr(const oracle_dasp_t *d, const char *name, int *len)
{
    FIELD_ARR(order_rid, d->rowA);
    FIELD_ARR(short_val, d->fill0_nnz_short); FIELD_ARR(short_cid, d->fill0_nnz_short);
    FIELD_ARR(long_val, d->fill0_nnz_long); FIELD_ARR(long_cid, d->fill0_nnz_long);
    FIELD_ARR(long_rpt_new, d->row_long + 1);
    FIELD_ARR(reg_val, d->fill0_nnz_reg); FIELD_ARR(reg_cid, d->fill0_nnz_reg);
    FIELD_ARR(block_ptr, d->blocknum + 1);
    FIELD_ARR(irreg_val, d->nnz_irreg); FIELD_ARR(irreg_cid, d->nnz_irreg);
    FIELD_ARR(irreg_rpt, d->row_block + 1);
    if (len) *len = -1;
    return NULL;
}

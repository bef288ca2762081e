// cli_rocsparse.cpp -- vendor comparator column, the analogue of the reference's cusparse_spmv_all
// (src/main_f64.cu:18-100: generic-API CSR SpMV, default algorithm, 100 warm-up + 1000 timed launches).
// Not part of libdasp_amd.so; prints one line next to dasp_bench's for context.
//   dasp_rocsparse <workload> [scale=1] [iters=200] [warmup=20]        (f64 only)
#include <hip/hip_runtime_api.h>
#include <rocsparse/rocsparse.h>

#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "../../include/dasp_amd.h"

#pragma clang diagnostic ignored "-Wdeprecated-declarations"
#define RS(x) do { rocsparse_status s_ = (x); if (s_ != rocsparse_status_success) { std::fprintf(stderr, "%s -> %d\n", #x, (int)s_); return 2; } } while (0)
#define HC(x) do { if ((x) != hipSuccess) { std::fprintf(stderr, "%s failed\n", #x); return 2; } } while (0)

int main(int argc, char **argv)
{
    if (argc < 2) { std::printf("usage: dasp_rocsparse <workload> [scale] [iters] [warmup]\n"); return 0; }
    const char *name = argv[1];
    const double scale = argc > 2 ? std::atof(argv[2]) : 1.0;
    const int iters = argc > 3 ? std::atoi(argv[3]) : 200, warmup = argc > 4 ? std::atoi(argv[4]) : 20;
    int rows, cols;
    if (dasp_synth_dims(name, scale, &rows, &cols)) return 1;
    std::vector<int> rp((size_t)rows + 1, 0);
    if (dasp_synth_row_lengths(name, scale, 0, rows, rp.data())) return 1;
    long long run = 0;
    for (int i = 0; i <= rows; ++i) { long long v = i < rows ? rp[i] : 0; rp[i] = (int)run; run += v; }
    const int nnz = rp[rows];
    std::vector<int> ci((size_t)nnz);
    if (dasp_synth_rows(name, scale, 0, rows, rp.data(), ci.data())) return 1;
    std::vector<double> val((size_t)nnz, 1.0), x((size_t)cols, 1.0), y((size_t)rows);
    int *drp, *dci; double *dv, *dx, *dy;
    HC(hipMalloc(&drp, sizeof(int) * ((size_t)rows + 1))); HC(hipMalloc(&dci, sizeof(int) * (size_t)nnz + 8));
    HC(hipMalloc(&dv, 8 * (size_t)nnz + 8)); HC(hipMalloc(&dx, 8 * (size_t)cols + 8)); HC(hipMalloc(&dy, 8 * (size_t)rows + 8));
    HC(hipMemcpy(drp, rp.data(), sizeof(int) * ((size_t)rows + 1), hipMemcpyHostToDevice));
    HC(hipMemcpy(dci, ci.data(), sizeof(int) * (size_t)nnz, hipMemcpyHostToDevice));
    HC(hipMemcpy(dv, val.data(), 8 * (size_t)nnz, hipMemcpyHostToDevice));
    HC(hipMemcpy(dx, x.data(), 8 * (size_t)cols, hipMemcpyHostToDevice));
    rocsparse_handle h; RS(rocsparse_create_handle(&h));
    rocsparse_spmat_descr A; rocsparse_dnvec_descr vx, vy;
    RS(rocsparse_create_csr_descr(&A, rows, cols, nnz, drp, dci, dv, rocsparse_indextype_i32, rocsparse_indextype_i32,
                                  rocsparse_index_base_zero, rocsparse_datatype_f64_r));
    RS(rocsparse_create_dnvec_descr(&vx, cols, dx, rocsparse_datatype_f64_r));
    RS(rocsparse_create_dnvec_descr(&vy, rows, dy, rocsparse_datatype_f64_r));
    const double alpha = 1.0, beta = 0.0;
    size_t bsz = 0; void *buf = nullptr;
    const auto p0 = std::chrono::steady_clock::now();
    RS(rocsparse_spmv(h, rocsparse_operation_none, &alpha, A, vx, &beta, vy, rocsparse_datatype_f64_r, rocsparse_spmv_alg_default,
                      rocsparse_spmv_stage_buffer_size, &bsz, nullptr));
    HC(hipMalloc(&buf, bsz + 8));
    RS(rocsparse_spmv(h, rocsparse_operation_none, &alpha, A, vx, &beta, vy, rocsparse_datatype_f64_r, rocsparse_spmv_alg_default,
                      rocsparse_spmv_stage_preprocess, &bsz, buf));
    HC(hipDeviceSynchronize());
    const double pre_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - p0).count();
    for (int i = 0; i < warmup; ++i)
        RS(rocsparse_spmv(h, rocsparse_operation_none, &alpha, A, vx, &beta, vy, rocsparse_datatype_f64_r, rocsparse_spmv_alg_default,
                          rocsparse_spmv_stage_compute, &bsz, buf));
    HC(hipDeviceSynchronize());
    const auto t0 = std::chrono::steady_clock::now();
    for (int i = 0; i < iters; ++i)
        RS(rocsparse_spmv(h, rocsparse_operation_none, &alpha, A, vx, &beta, vy, rocsparse_datatype_f64_r, rocsparse_spmv_alg_default,
                          rocsparse_spmv_stage_compute, &bsz, buf));
    HC(hipDeviceSynchronize());
    const double ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count() / iters;
    HC(hipMemcpy(y.data(), dy, 8 * (size_t)rows, hipMemcpyDeviceToHost));
    long long bad = 0;
    for (int i = 0; i < rows; ++i) bad += y[i] != (double)(rp[i + 1] - rp[i]);
    const double balg = (double)(nnz + cols + rows) * 8 + (double)nnz * 4 + (double)(rows + 1) * 4;
    std::printf("rocsparse(csr,default) %s scale=%g f64 rows=%d nnz=%d pre=%.1fms | %.4f ms %.1f GFLOP/s %.1f GB/s alg = %.3f of 8 TB/s | mismatches=%lld\n",
                name, scale, rows, nnz, pre_ms, ms, 2.0 * nnz / (ms * 1e6), balg / (ms * 1e6), balg / (ms * 1e6) / 8000.0, bad);
    return bad ? 3 : 0;
}

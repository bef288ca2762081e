// cli_rocsparse.cpp -- vendor comparator column, the analogue of the reference's cusparse_spmv_all
// (src/main_f64.cu:18-100: generic-API CSR SpMV, default algorithm, 100 warm-up + 1000 timed launches).
// Not part of libdasp_amd.so; prints one line next to dasp_bench's for context.
//   dasp_rocsparse <workload> [scale=1] [iters=200] [warmup=20] [compare=0]        (f64 only)
// compare=1: values and x seeded pseudo-random in (-1,1) instead of all ones, and the comparator's y checked against the DASP plan's y
// through order_rid -- the reference's verify_new (src/main_f64.cu:3-16: y_cusparse[order_rid[i]] vs y_dasp[i]), at 1e-12 relative to
// sum_j |a_ij x_j| instead of its 1e-5 absolute.  Exit code 3 on a mismatch.
#include <hip/hip_runtime_api.h>
#include <rocsparse/rocsparse.h>

#include <chrono>
#include <cmath>
#include <cstdint>
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
    const bool compare = argc > 5 && std::atoi(argv[5]) != 0;
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
    if (compare) {      // splitmix64 -> (-1, 1): any two entries differ, so a wrong column id or a swapped value shows
        uint64_t st = 0x9E3779B97F4A7C15ull;
        auto next = [&st] { uint64_t z = (st += 0x9E3779B97F4A7C15ull); z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull; z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
                            return (double)((z ^ (z >> 31)) >> 11) * (2.0 / 9007199254740992.0) - 1.0; };
        for (double &v : val) v = next();
        for (double &v : x) v = next();
    }
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
    double worst = 0;
    if (!compare) {
        for (int i = 0; i < rows; ++i) bad += y[i] != (double)(rp[i + 1] - rp[i]);
    } else {
        dasp_plan_t *plan = nullptr;
        std::vector<double> yd((size_t)rows);
        if (dasp_plan_create(&plan, 64, rows, cols, nnz, rp.data(), ci.data(), val.data(), nullptr) || dasp_plan_upload(plan) ||
            dasp_plan_spmv(plan, dx, dy, nullptr)) { std::fprintf(stderr, "dasp plan: %s\n", dasp_last_error()); return 2; }
        HC(hipDeviceSynchronize());
        HC(hipMemcpy(yd.data(), dy, 8 * (size_t)rows, hipMemcpyDeviceToHost));
        const int *order = dasp_plan_order(plan);
        for (int i = 0; i < rows; ++i) {
            const int r = order[i];
            double scale_r = 0;
            for (int j = rp[r]; j < rp[r + 1]; ++j) scale_r += std::fabs(val[(size_t)j] * x[(size_t)ci[(size_t)j]]);
            const double err = std::fabs(y[(size_t)r] - yd[(size_t)i]);
            const double rel = scale_r > 0 ? err / scale_r : err;
            if (!(rel <= 1e-12)) ++bad;
            if (rel > worst || rel != rel) worst = rel;
        }
        dasp_plan_destroy(plan);
    }
    const double balg = (double)(nnz + cols + rows) * 8 + (double)nnz * 4 + (double)(rows + 1) * 4;
    std::printf("rocsparse(csr,default) %s scale=%g f64 rows=%d nnz=%d pre=%.1fms | %.4f ms %.1f GFLOP/s %.1f GB/s alg = %.3f of 8 TB/s | mismatches=%lld%s\n",
                name, scale, rows, nnz, pre_ms, ms, 2.0 * nnz / (ms * 1e6), balg / (ms * 1e6), balg / (ms * 1e6) / 8000.0, bad,
                compare ? " (vs dasp_plan_spmv through order_rid, 1e-12 relative)" : "");
    if (compare) std::printf("compare: rows=%d max_rel_err=%.3e tol=1e-12\n", rows, worst);
    return bad ? 3 : 0;
}

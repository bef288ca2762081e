// cli_bench.cpp -- torch-free driver of the C ABI for profiling (rocprofv3 -- dasp_bench ...):
// builds the synthetic stand-in, runs the reference's timing protocol, prints one result line.
//   dasp_bench <workload> [scale=1] [precision=64] [iters=200] [warmup=20] [threshold=0.75] [long_piece=0] [x_window=0] [row_window=0] [cid16=0]
#include <hip/hip_runtime_api.h>

#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "../../include/dasp_amd.h"

static int s_panels(dasp_plan_t *p) { return dasp_plan_panel_count(p); }

#define CHECK(x) do { int rc_ = (x); if (rc_ != 0) { std::fprintf(stderr, "%s failed: %d (%s)\n", #x, rc_, dasp_last_error()); return 1; } } while (0)

int main(int argc, char **argv)
{
    if (argc < 2) { std::printf("usage: dasp_bench <workload> [scale] [precision] [iters] [warmup] [threshold] [long_piece] [x_window] [row_window] [cid16] [col_panels] [stream_policy] [slab_max_len] [piece_min_len] [x_window_hybrid] [chunk_pairs] [two_phase]\n"); return 0; }
    const char *name = argv[1];
    const double scale = argc > 2 ? std::atof(argv[2]) : 1.0;
    const int prec = argc > 3 ? std::atoi(argv[3]) : 64;
    const int iters = argc > 4 ? std::atoi(argv[4]) : 200;
    const int warmup = argc > 5 ? std::atoi(argv[5]) : 20;
    const double threshold = argc > 6 ? std::atof(argv[6]) : 0.75;
    const int long_piece = argc > 7 ? std::atoi(argv[7]) : 0;
    const int x_window = argc > 8 ? std::atoi(argv[8]) : 0;
    const int row_window = argc > 9 ? std::atoi(argv[9]) : 0;
    const int cid16 = argc > 10 ? std::atoi(argv[10]) : 0;
    const int col_panels = argc > 11 ? std::atoi(argv[11]) : 0;
    const int stream_policy = argc > 12 ? std::atoi(argv[12]) : 0;
    const int slab_max_len = argc > 13 ? std::atoi(argv[13]) : 0;
    const int piece_min_len = argc > 14 ? std::atoi(argv[14]) : 0;
    const int chunk_pairs = argc > 16 ? std::atoi(argv[16]) : 0;
    const int x_window_hybrid = argc > 15 ? std::atoi(argv[15]) : 0;
    const int two_phase = argc > 17 ? std::atoi(argv[17]) : 0;
    int rows, cols;
    CHECK(dasp_synth_dims(name, scale, &rows, &cols));
    std::vector<int> rp((size_t)rows + 1, 0);
    CHECK(dasp_synth_row_lengths(name, scale, 0, rows, rp.data()));
    long long run = 0;
    for (int i = 0; i <= rows; ++i) { long long v = i < rows ? rp[i] : 0; rp[i] = (int)run; run += v; }
    const int nnz = rp[rows];
    std::vector<int> ci((size_t)nnz);
    CHECK(dasp_synth_rows(name, scale, 0, rows, rp.data(), ci.data()));
    const size_t vb = prec == 64 ? 8 : 2;
    std::vector<char> val((size_t)nnz * vb);
    if (prec == 64) for (int i = 0; i < nnz; ++i) reinterpret_cast<double *>(val.data())[i] = 1.0;
    else for (int i = 0; i < nnz; ++i) reinterpret_cast<uint16_t *>(val.data())[i] = 0x3C00;
    dasp_options_t opt;
    dasp_options_default(&opt);
    opt.threshold = threshold; opt.long_piece = long_piece; opt.x_window = x_window; opt.row_window = row_window; opt.cid16 = cid16; opt.col_panels = col_panels; opt.stream_policy = stream_policy; opt.slab_max_len = slab_max_len; opt.piece_min_len = piece_min_len; opt.x_window_hybrid = x_window_hybrid; opt.chunk_pairs = chunk_pairs; opt.two_phase = two_phase;
    dasp_plan_t *plan = nullptr;
    CHECK(dasp_plan_create(&plan, prec, rows, cols, nnz, rp.data(), ci.data(), val.data(), &opt));
    CHECK(dasp_plan_upload(plan));
    std::vector<char> ones((size_t)cols * vb);
    if (prec == 64) for (int i = 0; i < cols; ++i) reinterpret_cast<double *>(ones.data())[i] = 1.0;
    else for (int i = 0; i < cols; ++i) reinterpret_cast<uint16_t *>(ones.data())[i] = 0x3C00;
    void *dX = nullptr, *dY = nullptr;
    if (hipMalloc(&dX, ones.size() + 8) != hipSuccess || hipMalloc(&dY, (size_t)rows * vb + 8) != hipSuccess) return 2;
    if (hipMemcpy(dX, ones.data(), ones.size(), hipMemcpyHostToDevice) != hipSuccess) return 2;
    double wall = 0, ev = 0;
    CHECK(dasp_plan_time(plan, dX, dY, nullptr, warmup, iters, &wall, &ev));
    double gwall = 0, gev = 0;
    CHECK(dasp_plan_time_graph(plan, dX, dY, nullptr, warmup, iters, iters < 50 ? iters : 50, &gwall, &gev));
    // exact check: y[i] == nnz(row order[i])
    std::vector<char> y((size_t)rows * vb);
    if (hipMemcpy(y.data(), dY, y.size(), hipMemcpyDeviceToHost) != hipSuccess) return 2;
    const int *order = dasp_plan_order(plan);
    long long bad = 0;
    if (prec == 64)
        for (int i = 0; i < rows; ++i) bad += reinterpret_cast<double *>(y.data())[i] != (double)(rp[order[i] + 1] - rp[order[i]]);
    else   // binary16 result: exact up to 2048, then within f16 rounding (1e-2 relative covers column panels too), +inf past 65504
        for (int i = 0; i < rows; ++i) {
            const uint16_t h = reinterpret_cast<uint16_t *>(y.data())[i];
            const int e = (h >> 10) & 31, f = h & 1023;
            const double got = (h >> 15) ? -1.0 : (e == 0 ? std::ldexp((double)f, -24) : (e == 31 ? (f ? NAN : INFINITY) : std::ldexp((double)(f | 1024), e - 25)));
            const double want = (double)(rp[order[i] + 1] - rp[order[i]]);
            const bool ok = want > 65504.0 ? (std::isinf(got) || std::fabs(got - want) <= 1e-2 * want)
                                           : (want <= 2048.0 && s_panels(plan) == 0 ? got == want : std::fabs(got - want) <= 1e-2 * (want > 1 ? want : 1));
            bad += !ok;
        }
    // second check, sensitive to the COLUMN ids (the all-ones product is not): x[j] = 1 + (j % 61) / 64 (f64; sums stay exact) or
    // 1 + (j % 5) / 4 (f16: exact products, f32 accumulation, 1e-2 relative on the rounded result), against a host CSR loop
    {
        std::vector<char> xv((size_t)cols * vb);
        std::vector<double> xd((size_t)cols);
        for (int j = 0; j < cols; ++j) {
            if (prec == 64) { xd[(size_t)j] = 1.0 + (double)(j % 61) / 64.0; reinterpret_cast<double *>(xv.data())[j] = xd[(size_t)j]; }
            else { const int q = j % 5; xd[(size_t)j] = 1.0 + q / 4.0; reinterpret_cast<uint16_t *>(xv.data())[j] = (uint16_t)(q == 4 ? 0x4000 : 0x3C00 + 0x100 * q); }
        }
        if (hipMemcpy(dX, xv.data(), xv.size(), hipMemcpyHostToDevice) != hipSuccess) return 2;
        CHECK(dasp_plan_spmv(plan, dX, dY, nullptr));
        if (hipDeviceSynchronize() != hipSuccess || hipMemcpy(y.data(), dY, y.size(), hipMemcpyDeviceToHost) != hipSuccess) return 2;
        for (int i = 0; i < rows; ++i) {
            const int r = order[i];
            double want = 0;
            for (int j = rp[r]; j < rp[r + 1]; ++j) want += xd[(size_t)ci[j]];
            if (prec == 64) bad += reinterpret_cast<double *>(y.data())[i] != want;
            else {
                const uint16_t h = reinterpret_cast<uint16_t *>(y.data())[i];
                const int e = (h >> 10) & 31, f = h & 1023;
                const double got = (h >> 15) ? -1.0 : (e == 0 ? std::ldexp((double)f, -24) : (e == 31 ? (f ? NAN : INFINITY) : std::ldexp((double)(f | 1024), e - 25)));
                bad += !(want > 65504.0 ? (std::isinf(got) || std::fabs(got - want) <= 1e-2 * want) : std::fabs(got - want) <= 1e-2 * (want > 1 ? want : 1));
            }
        }
    }
    dasp_stats_t s;
    dasp_plan_stats(plan, &s);
    // the same plan packed on the GPU from a device-resident CSR (dasp_plan_create_device): preprocessing time only
    double dev_pre = -1;
    {
        int *drp = nullptr, *dci = nullptr; void *dv = nullptr;
        if (hipMalloc(&drp, 4 * ((size_t)rows + 1)) == hipSuccess && hipMalloc(&dci, 4 * (size_t)nnz + 8) == hipSuccess && hipMalloc(&dv, vb * (size_t)nnz + 8) == hipSuccess &&
            hipMemcpy(drp, rp.data(), 4 * ((size_t)rows + 1), hipMemcpyHostToDevice) == hipSuccess && hipMemcpy(dci, ci.data(), 4 * (size_t)nnz, hipMemcpyHostToDevice) == hipSuccess &&
            hipMemcpy(dv, val.data(), vb * (size_t)nnz, hipMemcpyHostToDevice) == hipSuccess) {
            dasp_plan_t *dp = nullptr;
            dasp_options_t dopt = opt;
            dopt.col_panels = 1;                       // column panels are split on the host
            if (dasp_plan_create_device(&dp, prec, rows, cols, nnz, drp, dci, dv, &dopt) == 0) {
                dasp_stats_t ds; dasp_plan_stats(dp, &ds); dev_pre = ds.pre_ms;
                dasp_plan_destroy(dp);
            }
        }
        if (drp) (void)hipFree(drp); if (dci) (void)hipFree(dci); if (dv) (void)hipFree(dv);
    }
    const double balg = (double)s.data_origin1;
    std::printf("%s scale=%g f%d rows=%d nnz=%d long=%d med=%d fill0=%.4f pre=%.1fms devpre=%.1fms win=%d/%d lds=%dB c16=%d panels=%d pieces=%d tp=%d | %.4f ms (event %.4f) %.1f GFLOP/s %.1f GB/s alg = %.3f of 8 TB/s | graph: %.4f ms %.3f | mismatches=%lld\n",
                name, scale, prec, rows, nnz, s.row_long, s.row_block, s.rate_fill0, s.pre_ms, dev_pre, s.n_windows_lds, s.n_windows, s.lds_bytes, s.cid16_on, s.n_col_panels, s.med_rows_as_pieces, s.two_phase, wall, ev, 2.0 * nnz / (wall * 1e6),
                balg / (ev * 1e6), balg / (ev * 1e6) / 8000.0, gev, balg / (gev * 1e6) / 8000.0, bad);
    (void)hipFree(dX); (void)hipFree(dY);
    dasp_plan_destroy(plan);
    return bad ? 3 : 0;
}

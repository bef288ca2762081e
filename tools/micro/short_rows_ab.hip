// tools/micro/short_rows_ab.hip -- A/B of the two short-row designs on synthetic rows of one length L (1..4), f64 and f16:
//   (A) the product's layout: uniform-length slab [tile][k][R], a lane owns whole rows (2 rows f64 / 4 rows f16 per lane, 16-byte loads),
//       the segmented dot product needs no cross-lane step (dasp_amd/csrc/kernels.hip: short_rows);
//   (B) the north_star's wording: rows stored back to back in CSR order, one nonzero per lane, products summed per row with a
//       wavefront-segmented reduction on DPP row shifts (segments never straddle a wave: 64 / L whole rows per wave, the rest of the lanes idle
//       for L = 3), the segment heads store y.
// Same x (random columns over n_cols), same values; results compared.      hipcc --offload-arch=gfx950 -O3 short_rows_ab.hip -o short_rows_ab
// usage: short_rows_ab [rows=4000000] [n_cols=1000000]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <random>
typedef double f64x2 __attribute__((ext_vector_type(2)));
typedef int i32x2 __attribute__((ext_vector_type(2)));
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
typedef int i32x4 __attribute__((ext_vector_type(4)));

template <class T, int L> __global__ __launch_bounds__(256) void slab_kernel(const T *val, const int *cid, const T *x, T *y, int rows)
{
    constexpr int V = sizeof(T) == 8 ? 2 : 4, SR = 64 * V;
    const int lane = threadIdx.x & 63, tile = blockIdx.x * 4 + (threadIdx.x >> 6);
    if ((long long)tile * SR >= rows) return;
    const size_t base = (size_t)tile * L * SR + (size_t)V * lane;
    float sf[4] = {0, 0, 0, 0}; double sd[2] = {0, 0};
#pragma unroll
    for (int k = 0; k < L; ++k) {
        if constexpr (sizeof(T) == 8) {
            const f64x2 a = *reinterpret_cast<const f64x2 *>(val + base + (size_t)k * SR);
            const i32x2 c = *reinterpret_cast<const i32x2 *>(cid + base + (size_t)k * SR);
            sd[0] += a[0] * x[c[0]]; sd[1] += a[1] * x[c[1]];
        } else {
            const f16x4 a = *reinterpret_cast<const f16x4 *>(val + base + (size_t)k * SR);
            const i32x4 c = *reinterpret_cast<const i32x4 *>(cid + base + (size_t)k * SR);
#pragma unroll
            for (int v = 0; v < 4; ++v) sf[v] += (float)a[v] * (float)x[c[v]];
        }
    }
    const int r0 = tile * SR + V * lane;
#pragma unroll
    for (int v = 0; v < V; ++v) if (r0 + v < rows) y[r0 + v] = sizeof(T) == 8 ? (T)sd[v & 1] : (T)sf[v];
}

template <int CTRL> __device__ __forceinline__ float dpp_f(float v) { return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xf, 0xf, true)); }
template <int CTRL> __device__ __forceinline__ double dpp_d(double v)
{
    int lo = __builtin_amdgcn_update_dpp(0, __double2loint(v), CTRL, 0xf, 0xf, true), hi = __builtin_amdgcn_update_dpp(0, __double2hiint(v), CTRL, 0xf, 0xf, true);
    return __hiloint2double(hi, lo);
}
// one nonzero per lane; a wave holds RPW = 64 / L whole rows (lanes >= RPW * L idle); rows never straddle a 16-lane DPP row for L = 1, 2, 4;
// for L = 3 the rows are laid out 5 per DPP row (15 lanes, lane 15 of each DPP row idle) so that the shifts stay inside a row of 16 lanes
template <class T, int L> __global__ __launch_bounds__(256) void seg_kernel(const T *val, const int *cid, const T *x, T *y, int rows)
{
    constexpr int PER16 = 16 / L, RPW = 4 * PER16;                 // rows per 16 lanes / per wave
    const int lane = threadIdx.x & 63, wave = blockIdx.x * 4 + (threadIdx.x >> 6);
    const int sub = lane & 15, q = lane >> 4;
    const int rloc = sub / L, k = sub % L;
    const long long row = (long long)wave * RPW + q * PER16 + rloc;
    const bool live = rloc < PER16 && row < rows;
    using P = typename std::conditional<sizeof(T) == 8, double, float>::type;
    P p = 0;
    if (live) { const size_t e = (size_t)row * L + k; p = (P)val[e] * (P)x[cid[e]]; }
    // segmented sum towards the segment head (k == 0): add the value 1 lane to the right, then 2 lanes to the right (row_shl: lane i reads lane i + n)
    if constexpr (L >= 2) {
        P t;
        if constexpr (sizeof(T) == 8) t = dpp_d<0x101>(p); else t = dpp_f<0x101>(p);
        if (k + 1 < L) p += t;
    }
    if constexpr (L >= 3) {
        P t;
        if constexpr (sizeof(T) == 8) t = dpp_d<0x102>(p); else t = dpp_f<0x102>(p);
        if (k == 0 && L >= 3) p += t;                               // head + (elements 2, 3 already folded into lane head + 2)
    }
    if (live && k == 0) y[row] = (T)p;
}

template <class T> void run(const char *name, int rows, int n_cols)
{
    constexpr int V = sizeof(T) == 8 ? 2 : 4, SR = 64 * V;
    std::mt19937_64 rng(7);
    std::vector<T> hx((size_t)n_cols);
    for (auto &v : hx) v = (T)(0.5 + (rng() % 1024) / 1024.0);
    T *dx; hipMalloc(&dx, hx.size() * sizeof(T)); hipMemcpy(dx, hx.data(), hx.size() * sizeof(T), hipMemcpyHostToDevice);
    T *ya, *yb; hipMalloc(&ya, (size_t)rows * sizeof(T)); hipMalloc(&yb, (size_t)rows * sizeof(T));
    for (int L = 1; L <= 4; ++L) {
        const size_t tiles = ((size_t)rows + SR - 1) / SR, nslab = tiles * L * SR, ncsr = (size_t)rows * L;
        std::vector<T> va(nslab, (T)0), vb(ncsr); std::vector<int> ca(nslab, 0), cb(ncsr);
        for (size_t r = 0; r < (size_t)rows; ++r)
            for (int k = 0; k < L; ++k) {
                const T v = (T)(0.5 + (rng() % 512) / 512.0); const int c = (int)(rng() % (unsigned long long)n_cols);
                vb[r * L + k] = v; cb[r * L + k] = c;
                const size_t at = (r / SR) * L * SR + (size_t)k * SR + r % SR; va[at] = v; ca[at] = c;
            }
        T *dva, *dvb; int *dca, *dcb;
        hipMalloc(&dva, nslab * sizeof(T)); hipMalloc(&dca, nslab * 4); hipMalloc(&dvb, ncsr * sizeof(T)); hipMalloc(&dcb, ncsr * 4);
        hipMemcpy(dva, va.data(), nslab * sizeof(T), hipMemcpyHostToDevice); hipMemcpy(dca, ca.data(), nslab * 4, hipMemcpyHostToDevice);
        hipMemcpy(dvb, vb.data(), ncsr * sizeof(T), hipMemcpyHostToDevice); hipMemcpy(dcb, cb.data(), ncsr * 4, hipMemcpyHostToDevice);
        const int ga = (int)((tiles + 3) / 4);
        const int rpw = 4 * (16 / L), gb = (int)(((size_t)rows + rpw - 1) / rpw + 3) / 4;
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        float ms[2] = {0, 0};
        for (int which = 0; which < 2; ++which) {
            auto launch = [&] {
#define GO(LL) if (which == 0) hipLaunchKernelGGL((slab_kernel<T, LL>), dim3(ga), dim3(256), 0, 0, dva, dca, dx, ya, rows); else hipLaunchKernelGGL((seg_kernel<T, LL>), dim3(gb), dim3(256), 0, 0, dvb, dcb, dx, yb, rows);
                switch (L) { case 1: GO(1) break; case 2: GO(2) break; case 3: GO(3) break; default: GO(4) }
#undef GO
            };
            for (int i = 0; i < 5; ++i) launch();
            hipEventRecord(e0, 0);
            for (int i = 0; i < 50; ++i) launch();
            hipEventRecord(e1, 0); hipEventSynchronize(e1);
            hipEventElapsedTime(&ms[which], e0, e1); ms[which] /= 50;
        }
        std::vector<T> ha((size_t)rows), hb((size_t)rows);
        hipMemcpy(ha.data(), ya, (size_t)rows * sizeof(T), hipMemcpyDeviceToHost); hipMemcpy(hb.data(), yb, (size_t)rows * sizeof(T), hipMemcpyDeviceToHost);
        size_t bad = 0;
        for (size_t r = 0; r < (size_t)rows; ++r) { const double d = (double)ha[r] - (double)hb[r], s = (double)ha[r]; if (d > 1e-2 * s || -d > 1e-2 * s) ++bad; }
        const double bytes = (double)rows * L * (sizeof(T) + 4) + (double)rows * sizeof(T);
        printf("%s rows of %d: slab (lane owns rows) %.4f ms = %.2f TB/s | wave-segmented DPP %.4f ms = %.2f TB/s | mismatches %zu\n", name, L, ms[0], bytes / ms[0] / 1e9,
               ms[1], bytes / ms[1] / 1e9, bad);
        hipFree(dva); hipFree(dca); hipFree(dvb); hipFree(dcb);
    }
    hipFree(dx); hipFree(ya); hipFree(yb);
}
int main(int argc, char **argv)
{
    const int rows = argc > 1 ? atoi(argv[1]) : 4000000, n_cols = argc > 2 ? atoi(argv[2]) : 1000000;
    run<double>("f64", rows, n_cols);
    run<_Float16>("f16", rows, n_cols);
    return 0;
}

// tools/proto/twophase.hip -- PROTOTYPE: a gather-free two-phase SpMV for graph matrices in f16 (BASELINE config 4).
//
// Why: every kernel that gathers x per nonzero sits at ~0.8 L1 misses per nonzero on these matrices (profiles/r04_row_tiles.md section 10); the
// DASP layout cannot get under that.  Here no gather ever leaves the CU:
//   phase 1 (tp_expand): column blocks of CB columns.  A workgroup stages its block's slice of x in LDS (coalesced 16-byte loads) and streams the
//            block's nonzeros as u16 LOCAL column ids, 64-element segments in (column block, row block) tile order; it writes xs[k] = x[col[k]]
//            -- the x value every nonzero needs -- as a contiguous f16 stream laid out in (row block, column block) tile order, one 128-byte
//            segment at a time (dst_seg[] says where a segment goes).
//   phase 2 (tp_reduce): row blocks of RB rows.  A workgroup keeps its slice of y in LDS as f32, streams (value, u16 LOCAL row id, xs) -- all three
//            contiguous -- and accumulates value * xs with LDS float atomics; then y is written once, coalesced.
// Every byte is streamed: 2 (local column) + 2 (xs written) + 2 (value) + 2 (local row) + 2 (xs read) = 10 B per nonzero against B_alg's 6.
// The arithmetic is the direct kernel's: f16 x f16 products accumulated in f32 (the order of the additions is not fixed: LDS atomics).
//   hipcc -O3 --offload-arch=gfx950 -shared -fPIC -o dasp_amd/variants/proto/libtwophase.so tools/proto/twophase.hip
#include <hip/hip_runtime.h>
#include <cstdint>

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef unsigned short u16x8 __attribute__((ext_vector_type(8)));
typedef int i32x4 __attribute__((ext_vector_type(4)));

// grid = n_cb * splits workgroups of 512 threads; dynamic LDS = CB * 2 bytes
__global__ __launch_bounds__(512) void tp_expand(const unsigned short *__restrict__ lcol, const int *__restrict__ dst_seg, const _Float16 *__restrict__ x,
                                                 _Float16 *__restrict__ xs, const int *__restrict__ cb_seg0, int CB, int n, int splits)
{
    extern __shared__ __attribute__((aligned(16))) char lds_raw[];
    _Float16 *xl = reinterpret_cast<_Float16 *>(lds_raw);
    const int c = blockIdx.x / splits, s = blockIdx.x % splits;
    const int c0 = c * CB, len = min(CB, n - c0);
    // the slice of x: 16-byte loads where whole (x + c0 is 16-byte aligned: CB is a multiple of 8)
    for (int i = threadIdx.x * 8; i < len; i += 512 * 8) {
        if (i + 8 <= len) *reinterpret_cast<i32x4 *>(xl + i) = *reinterpret_cast<const i32x4 *>(x + c0 + i);
        else for (int j = i; j < len; ++j) xl[j] = x[c0 + j];
    }
    __syncthreads();
    const int S0 = cb_seg0[c], S1 = cb_seg0[c + 1];
    const int per = ((S1 - S0 + splits - 1) / splits + 7) & ~7;          // segments per split, whole groups of 8 (one wave iteration)
    const int a = S0 + s * per, b = min(S1, a + per);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int sub = lane >> 3, off = (lane & 7) * 8;
#pragma unroll 2
    for (int g = a + wave * 8; g < b; g += 8 * 8) {
        const int seg = g + sub;
        if (seg < b) {
            const u16x8 lc = __builtin_nontemporal_load(reinterpret_cast<const u16x8 *>(lcol + (size_t)seg * 64 + off));
            const int d = dst_seg[seg];
            f16x8 o;
#pragma unroll
            for (int j = 0; j < 8; ++j) o[j] = xl[lc[j]];
            *reinterpret_cast<f16x8 *>(xs + (size_t)d * 64 + off) = o;
        }
    }
}

// grid = n_rb workgroups of THREADS threads; dynamic LDS = RB * 4 bytes
// MODE (timing experiments; only 0 is correct): 0 = LDS float atomics, 1 = plain read-add-write (races between waves), 2 = integer atomics on the bits,
// 3 = no LDS operation at all (the streaming floor)
template <int THREADS, int MODE = 0>
__global__ __launch_bounds__(THREADS) void tp_reduce(const _Float16 *__restrict__ val, const unsigned short *__restrict__ lrow, const _Float16 *__restrict__ xs,
                                                     const int *__restrict__ rb_seg0, _Float16 *__restrict__ y, int RB, int m)
{
    extern __shared__ __attribute__((aligned(16))) char lds_raw[];
    float *yl = reinterpret_cast<float *>(lds_raw);
    double *yd = reinterpret_cast<double *>(lds_raw);
    const int r = blockIdx.x, r0 = r * RB, len = min(RB, m - r0);
    if constexpr (MODE == 4 || MODE == 6) { for (int i = threadIdx.x; i < RB; i += THREADS) yd[i] = 0.0; }
    else for (int i = threadIdx.x; i < RB; i += THREADS) yl[i] = 0.0f;
    __syncthreads();
    const int S0 = rb_seg0[r], S1 = rb_seg0[r + 1];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int sub = lane >> 3, off = (lane & 7) * 8;
    float acc = 0.0f;
#pragma unroll 2
    for (int g = S0 + wave * 8; g < S1; g += (THREADS / 64) * 8) {
        const int seg = g + sub;
        if (seg < S1) {
            const size_t at = (size_t)seg * 64 + off;
            const f16x8 v = __builtin_nontemporal_load(reinterpret_cast<const f16x8 *>(val + at));
            const u16x8 lr = __builtin_nontemporal_load(reinterpret_cast<const u16x8 *>(lrow + at));
            const f16x8 xv = __builtin_nontemporal_load(reinterpret_cast<const f16x8 *>(xs + at));
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const float p = (float)v[j] * (float)xv[j];
                if constexpr (MODE == 0) { if (v[j] != (_Float16)0) __hip_atomic_fetch_add(&yl[lr[j]], p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }      // (pads carry value 0)
                else if constexpr (MODE == 1) yl[lr[j]] += p;
                else if constexpr (MODE == 2) __hip_atomic_fetch_add(reinterpret_cast<unsigned *>(yl) + lr[j], __float_as_uint(p), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                else if constexpr (MODE == 4) { if (v[j] != (_Float16)0) __hip_atomic_fetch_add(yd + lr[j], (double)p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }      // f64 accumulators: ds_add_f64 runs ~4x the rate of ds_add_f32 on gfx950
                else if constexpr (MODE == 6) __hip_atomic_fetch_add(reinterpret_cast<unsigned long long *>(yd) + lr[j], (unsigned long long)(long long)(p * 1048576.0f), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);      // (timing: 64-bit integer atomics)
                else if constexpr (MODE == 5) { typedef _Float16 h2 __attribute__((ext_vector_type(2))); const h2 q = {(_Float16)p, (_Float16)0};
                                                __builtin_amdgcn_ds_atomic_fadd_v2f16((__attribute__((address_space(3))) h2 *)(yl + lr[j]), q); }
                else acc += p;
            }
        }
    }
    if constexpr (MODE == 3) yl[threadIdx.x] = acc;
    __syncthreads();
    if constexpr (MODE == 4) { for (int i = threadIdx.x; i < len; i += THREADS) y[r0 + i] = (_Float16)(float)yd[i]; }
    else for (int i = threadIdx.x; i < len; i += THREADS) y[r0 + i] = (_Float16)yl[i];
}

extern "C" int tp_set_lds(int cb_bytes, int rb_bytes)
{
    if (hipFuncSetAttribute(reinterpret_cast<const void *>(&tp_expand), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess) return 1;
    if (hipFuncSetAttribute(reinterpret_cast<const void *>(&tp_reduce<256>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess) return 2;
    if (hipFuncSetAttribute(reinterpret_cast<const void *>(&tp_reduce<512>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess) return 3;
    (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&tp_reduce<256, 1>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&tp_reduce<256, 2>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&tp_reduce<256, 3>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&tp_reduce<256, 4>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&tp_reduce<512, 4>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&tp_reduce<512, 6>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&tp_reduce<256, 6>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&tp_reduce<512, 1>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&tp_reduce<512, 2>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&tp_reduce<512, 3>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&tp_reduce<512, 5>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&tp_reduce<256, 5>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    return 0;
}
extern "C" int tp_phase1(const void *lcol, const void *dst_seg, const void *x, void *xs, const void *cb_seg0, int CB, int n, int n_cb, int splits, void *stream)
{
    hipLaunchKernelGGL(tp_expand, dim3(n_cb * splits), dim3(512), (size_t)CB * 2, static_cast<hipStream_t>(stream), static_cast<const unsigned short *>(lcol),
                       static_cast<const int *>(dst_seg), static_cast<const _Float16 *>(x), static_cast<_Float16 *>(xs), static_cast<const int *>(cb_seg0), CB, n, splits);
    return (int)hipGetLastError();
}
extern "C" int tp_phase2(const void *val, const void *lrow, const void *xs, const void *rb_seg0, void *y, int RB, int m, int n_rb, int threads, void *stream)
{
    const int mode = threads / 1000; threads %= 1000;
#define TP_M(M) if (mode == M && threads == 512) { hipLaunchKernelGGL((tp_reduce<512, M>), dim3(n_rb), dim3(512), (size_t)RB * (M == 4 || M == 6 ? 8 : 4), static_cast<hipStream_t>(stream), static_cast<const _Float16 *>(val), \
                           static_cast<const unsigned short *>(lrow), static_cast<const _Float16 *>(xs), static_cast<const int *>(rb_seg0), static_cast<_Float16 *>(y), RB, m); return (int)hipGetLastError(); } \
    if (mode == M) { hipLaunchKernelGGL((tp_reduce<256, M>), dim3(n_rb), dim3(256), (size_t)RB * (M == 4 || M == 6 ? 8 : 4), static_cast<hipStream_t>(stream), static_cast<const _Float16 *>(val), \
                           static_cast<const unsigned short *>(lrow), static_cast<const _Float16 *>(xs), static_cast<const int *>(rb_seg0), static_cast<_Float16 *>(y), RB, m); return (int)hipGetLastError(); }
    TP_M(1) TP_M(2) TP_M(3) TP_M(4) TP_M(5) TP_M(6)
#undef TP_M
    if (threads == 512)
        hipLaunchKernelGGL(tp_reduce<512>, dim3(n_rb), dim3(512), (size_t)RB * 4, static_cast<hipStream_t>(stream), static_cast<const _Float16 *>(val),
                           static_cast<const unsigned short *>(lrow), static_cast<const _Float16 *>(xs), static_cast<const int *>(rb_seg0), static_cast<_Float16 *>(y), RB, m);
    else
        hipLaunchKernelGGL(tp_reduce<256>, dim3(n_rb), dim3(256), (size_t)RB * 4, static_cast<hipStream_t>(stream), static_cast<const _Float16 *>(val),
                           static_cast<const unsigned short *>(lrow), static_cast<const _Float16 *>(xs), static_cast<const int *>(rb_seg0), static_cast<_Float16 *>(y), RB, m);
    return (int)hipGetLastError();
}

// kernels.hip -- the gfx950 (CDNA4, wave64) SpMV kernels of a DASP plan, their launchers and the timing protocol.
//
// One fused launch covers the three row categories by workgroup-index range, as the
// reference's dasp_spmv2 does (src/dasp_f64.h:77-484, src/dasp_f16.h:133-590):
//   [ long pieces | medium blocks | short tiles ]      256 threads = 4 waves, one unit per wave
// followed by long_reduce only when some long row was cut into several pieces
// (the reference's longPart_sum, src/dasp_f64.h:53-75).
//
// The device code they are made of (fragments, block sources, the software pipeline, the three row categories) is spmv_device.hpp;
// the plan's way onto the device is upload.cpp; the multi-GPU step kernels are mgstep.hip.
//
// dasp_spmv_kernel<T, NT, C16, WIN>:
//   NT   streamed tiles with non-temporal loads (the reference's ld.global.cs "bypass" kernel) or plain loads
//   C16  regular medium tiles carry u16 column offsets from a per-chunk base (10 instead of 12 bytes per f64 nonzero)
//   WIN  windowed mode: one window of rows per 1024-thread workgroup, its span of x staged once in LDS
//        (dynamic LDS), every gather of the window served from LDS; y through med_dst
// DevArgs::acc turns every store of a row's result into y += (dasp_plan_spmv_acc); a plan split into column panels runs one
// such launch per panel into a partial buffer and dasp_panel_sum_kernel adds the partials.
// Rejected variants (XCD-contiguous ranges, persistent f64 grids, nt / sc1 gathers, vector tail loads, ...) are recorded in
// DESIGN.md section 4.4 and are not kept in this file; tools/ab.sh compares builds of two git revisions instead.
#include <hip/hip_runtime.h>

#include <chrono>
#include <thread>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "plan.hpp"
#include "device.hpp"

#include "spmv_device.hpp"

namespace dasp {
#ifdef DASP_STAMPS
extern int g_stamp_launch;
#endif

// MW (f64, no windows): 7 = the build held to 72 registers = 7 waves per SIMD, for plans of one-shot blocks (DevicePlan::seven_waves)
// L16 (r6): the build that reads the 16-bit ids of narrow long pieces (plan.hpp long_cid16) -- instantiations of their own, launched for plans in which those pieces
// matter (DevicePlan::long16): compiled into the ordinary builds, the third piece source cost every plan 1-3 % (HV15R-unstructured 507 -> 523 us, powerlaw_1M 374 -> 379,
// webbase-1M f64 28.9 -> 29.3, same box) for code that only long-row matrices run
template <class T, bool NT, bool C16, bool WIN, bool C8 = false, int MW = 0, bool L16 = false>
__global__ __launch_bounds__(WIN ? 1024 : 256, WIN ? kMinWavesWin : MW ? MW : kMinWavesPlain) void dasp_spmv_kernel(CallArgs c)
{
    extern __shared__ __attribute__((aligned(16))) char lds_raw[];
#ifdef DASP_STAMPS
    Stamps st; st.begin(c.stamp_launch);
#else
    Stamps st; st.begin(0);
#endif
    const DevArgs a = load_args(c);
    static_assert(!L16 || (!WIN && !C8 && MW == 0), "the 16-bit long ids ride in the plain build only");
    spmv_body<T, NT, C16, WIN, C8, false, WIN, L16>(a, lds_raw, blockIdx.x, st);
#ifdef DASP_STAMPS
    st.finish(a.wpw);
#endif
}

// a column panel with row tiles (Plan::rt_*): the non-windowed kernel + the tiles' workgroup range, dynamic LDS = 4 waves x 64 x rt_max products
// (f16: held to 72 registers = 7 waves per SIMD like the kernel without tiles -- 75 otherwise; ljournal-2008 0.4522 -> 0.4443 ms)
template <class T, bool NT, bool C16>
__global__ __launch_bounds__(256, sizeof(T) == 2 ? 7 : kMinWavesPlain) void dasp_spmv_rt_kernel(CallArgs c)
{
    extern __shared__ __attribute__((aligned(16))) char lds_raw[];
#ifdef DASP_STAMPS
    Stamps st; st.begin(c.stamp_launch);
#else
    Stamps st; st.begin(0);
#endif
    const DevArgs a = load_args(c);
    spmv_body<T, NT, C16, false, false, true>(a, lds_raw, blockIdx.x, st);
}

// the windowed kernel for plans with at most one window workgroup per CU (n_windows <= CUs: cop20k_A's 212): nothing is gained by
// holding it to 64 registers for a second resident workgroup that does not exist, and at 128 it needs no scratch.
// (r3, measured and NOT kept: a wave requesting the row tables, tiles and tails of both its blocks BEFORE the x copy and the barrier,
// so that only LDS gathers and MFMAs are left behind it -- cop20k_A 10.4 -> 10.9 us.  What bounds such a workgroup is not its chain
// of latencies but its CU's memory-level parallelism: ~200 KB per CU through 64 outstanding L1 misses of ~740 cycles)
template <class T, bool C16>
__global__ __launch_bounds__(1024, 4) void dasp_spmv_win1_kernel(CallArgs c)
{
    extern __shared__ __attribute__((aligned(16))) char lds_raw[];
#ifdef DASP_STAMPS
    Stamps st; st.begin(c.stamp_launch);
#else
    Stamps st; st.begin(0);
#endif
    const DevArgs a = load_args(c);
    // tiles, tails and ids stream with non-temporal loads although the matrix is cache-sized (r6, same box: cop20k_A 9.24 -> 8.91 us, x2 16.0 -> 15.4; every 4th window
    // only: 9.53): what the XCD's L2 then holds is the x spans neighbouring windows share and the row tables, and nothing of the once-per-launch stream pushes them out
    spmv_body<T, true, C16, true, false>(a, lds_raw, blockIdx.x, st);
#ifdef DASP_STAMPS
    st.finish(a.wpw);
#endif
}

// ---- all column panels of a plan in ONE launch (r5): the panels' grids back to back in one grid -- panel = a range of blockIdx.x, as the
// row categories are inside a panel (and as the reference ranges its categories: dasp_f64.h:1194-1212) -- so that the stream carries one launch + one
// stage 2 + the sum instead of 2 P + 1 launches with P tails.  Every panel keeps its own device-resident DevArgs; the kernel picks the pointer by range
// (constant indices + scalar selects: a run-time index into the kernarg arrays would copy them to scratch).
constexpr int kMaxMergedPanels = 8;
struct PanelCall {
    const DevArgs *plan[kMaxMergedPanels];
    int wg_end[kMaxMergedPanels];          // end of panel k's workgroup range (spmv launch) / of its stage-2 range (long_reduce launch)
    const void *x; char *part; size_t stride_bytes; int np;
};
__device__ __forceinline__ CallArgs panel_of(const PanelCall &c, int &wg)
{
    int k = 0, first = 0;
    const DevArgs *p = c.plan[0];
#pragma unroll
    for (int i = 1; i < kMaxMergedPanels; ++i) {
        const bool in = i < c.np && (int)blockIdx.x >= c.wg_end[i - 1];
        k = in ? i : k; first = in ? c.wg_end[i - 1] : first; p = in ? c.plan[i] : p;
    }
    wg = (int)blockIdx.x - first;
#ifdef DASP_STAMPS
    return CallArgs{p, c.x, c.part + (size_t)k * c.stride_bytes, 0, 0, -1};
#else
    return CallArgs{p, c.x, c.part + (size_t)k * c.stride_bytes, 0, 0};
#endif
}
template <class T, bool NT, bool C16>
__global__ __launch_bounds__(256, sizeof(T) == 2 ? 7 : kMinWavesPlain) void dasp_spmv_panels_kernel(PanelCall c)
{
    extern __shared__ __attribute__((aligned(16))) char lds_raw[];
    int wg;
    const CallArgs ca = panel_of(c, wg);
    Stamps st; st.begin(0);
    const DevArgs a = load_args(ca);
    spmv_body<T, NT, C16, false, false, true>(a, lds_raw, wg, st);
}

// stage 2 for long rows cut into several pieces (reference: longPart_sum, dasp_f64.h:53-75)
template <class T>
__global__ __launch_bounds__(256) void dasp_long_reduce_kernel(CallArgs c)
{
    const DevArgs a = load_args(c);
    using part_t = typename Tr<T>::part_t;
    // one WORKGROUP per row, four loads in flight per thread (late r5): one wave walking a row's partial sums 64 at a time took as long as the main kernel on a row of 50 M nonzeros
    // (51 200 partial sums: 354 us for 1 GB, tools/scratch/shape_probe.py).  A fixed order of additions: thread t takes t, t + 256, ...; lanes, then the four waves, are combined in order
    __shared__ part_t wsum[kWavesPerWG];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int i = blockIdx.x;
    const int q0 = a.multi_ptr[i], q1 = a.multi_ptr[i + 1];
    const part_t *part = static_cast<const part_t *>(a.partial);
    part_t s0 = 0, s1 = 0, s2 = 0, s3 = 0;
    int q = q0 + (int)threadIdx.x;
    for (; q + 768 < q1; q += 1024) { s0 += part[q]; s1 += part[q + 256]; s2 += part[q + 512]; s3 += part[q + 768]; }
    for (; q < q1; q += 256) s0 += part[q];
    const part_t s = wave_sum((s0 + s1) + (s2 + s3));
    if (lane == 0) wsum[wave] = s;
    __syncthreads();
    if (threadIdx.x == 0) put_y<T>(a, a.multi_dst[i], (wsum[0] + wsum[1]) + (wsum[2] + wsum[3]));
}
template <class T>
__global__ __launch_bounds__(256) void dasp_long_reduce_panels_kernel(PanelCall c)
{
    using part_t = typename Tr<T>::part_t;
    int wg;
    const CallArgs ca = panel_of(c, wg);
    const DevArgs a = load_args(ca);
    const int lane = threadIdx.x & 63;
    const int i = wg * kWavesPerWG + (threadIdx.x >> 6);
    if (i >= a.n_multi) return;
    const int q0 = a.multi_ptr[i], q1 = a.multi_ptr[i + 1];
    const part_t *part = static_cast<const part_t *>(a.partial);
    part_t s = 0;
    for (int q = q0 + lane; q < q1; q += kWave) s += part[q];
    s = wave_sum(s);
    if (lane == 0) put_y<T>(a, a.multi_dst[i], s);
}

// column panels (dasp_options_t::col_panels): y[i] = sum over the panels of part[k][i].  Streaming, V elements (16 bytes)
// per thread; the partial results are read once, so they bypass the caches.
template <class T, int V>
__global__ __launch_bounds__(256) void dasp_panel_sum_kernel(const T *__restrict__ part, size_t stride, int np, T *__restrict__ y, int m, int accum)
{
    using Acc = typename Tr<T>::part_t;
    const long long i0 = ((long long)blockIdx.x * 256 + threadIdx.x) * V;
    if (i0 >= m) return;
    if (V > 1 && i0 + V <= m) {
        typedef T vec_t __attribute__((ext_vector_type(V)));
        Acc acc[V];
#pragma unroll
        for (int j = 0; j < V; ++j) acc[j] = (Acc)0;
        if (accum) {
            const vec_t o = *reinterpret_cast<const vec_t *>(y + i0);
#pragma unroll
            for (int j = 0; j < V; ++j) acc[j] = (Acc)o[j];
        }
        for (int k = 0; k < np; ++k) {
            const vec_t v = __builtin_nontemporal_load(reinterpret_cast<const vec_t *>(part + (size_t)k * stride + i0));
#pragma unroll
            for (int j = 0; j < V; ++j) acc[j] += (Acc)v[j];
        }
        vec_t o;
#pragma unroll
        for (int j = 0; j < V; ++j) o[j] = (T)acc[j];
        *reinterpret_cast<vec_t *>(y + i0) = o;
    } else {
        for (long long i = i0; i < m && i < i0 + V; ++i) {
            Acc acc = accum ? (Acc)y[i] : (Acc)0;
            for (int k = 0; k < np; ++k) acc += (Acc)part[(size_t)k * stride + i];
            y[i] = (T)acc;
        }
    }
}


// ------------------------------------------------------------------ two-phase (gather-free) form: plan.hpp struct TwoPhase, DESIGN.md section 4.7
// Phase 1: one workgroup per unit (<= kTpUnitSegs segments of ONE column block, CB-major).  The block's slice of x is staged in LDS with
// coalesced 16-byte loads; then every lane takes 8 consecutive local column ids (one 16-byte load), reads their x values from LDS and stores
// them (one 16-byte store) where phase 2 will stream them.  8 lanes = one 128-byte segment; dst[] maps a CB-major segment to its RB-major place.
typedef unsigned short tp_u16x8 __attribute__((ext_vector_type(8)));
template <class T>
__global__ __launch_bounds__(512) void dasp_tp_expand_kernel(TpDev a, const T *__restrict__ x)
{
    extern __shared__ __attribute__((aligned(16))) char lds_raw[];
    typedef T vec8 __attribute__((ext_vector_type(8)));
    T *xl = reinterpret_cast<T *>(lds_raw);
    const int u = blockIdx.x;
    const int c = a.unit[3 * u], s0 = a.unit[3 * u + 1], s1 = a.unit[3 * u + 2];
    const int c0 = c * a.cb, len = min(a.cb, a.xlen - c0);
    if ((reinterpret_cast<uintptr_t>(x) & 15) == 0) {          // (c0 is a multiple of 8 elements)
        for (int i = threadIdx.x * 8; i < len; i += 512 * 8) {
            if (i + 8 <= len) *reinterpret_cast<vec8 *>(xl + i) = *reinterpret_cast<const vec8 *>(x + c0 + i);
            else for (int j = i; j < len; ++j) xl[j] = x[c0 + j];
        }
    } else for (int i = threadIdx.x; i < len; i += 512) xl[i] = x[c0 + i];
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    constexpr int LPS = kTpSeg / 8, SPW = 64 / LPS;          // lanes per segment (8 elements each), segments per wave iteration
    const int sub = lane / LPS, off = (lane % LPS) * 8;
    T *xs = static_cast<T *>(a.xs);
#pragma unroll 2
    for (int g = s0 + wave * SPW; g < s1; g += 8 * SPW) {
        const int seg = g + sub;
        if (seg < s1) {
            const tp_u16x8 lc = __builtin_nontemporal_load(reinterpret_cast<const tp_u16x8 *>(a.lcol + (size_t)seg * kTpSeg + off));
            const int d = a.dst[seg];
            vec8 o;
#pragma unroll
            for (int j = 0; j < 8; ++j) o[j] = xl[lc[j]];
            *reinterpret_cast<vec8 *>(xs + (size_t)d * kTpSeg + off) = o;
        }
    }
}

// Phase 2: one workgroup per row block.  Its <= rb_max output positions are f64 accumulators in LDS; the block's (value, local row, xs) triples are
// three contiguous streams read with 16-byte loads; every product (f16 x f16, exact in f32) is added with ds_add_f64 -- on gfx950 LDS f64 atomics
// run ~4x the rate of f32 ones (profiles/r05_two_phase.md: 0.153 against 0.559 ms for ljournal-2008), and the sum is exact to well below the f16
// result's rounding.  Then y is stored once, coalesced (acc: y += sum).
template <class T>
__global__ __launch_bounds__(512) void dasp_tp_reduce_kernel(TpDev a, T *__restrict__ y, int acc)
{
    extern __shared__ __attribute__((aligned(16))) char lds_raw[];
    typedef T vec8 __attribute__((ext_vector_type(8)));
    double *yl = reinterpret_cast<double *>(lds_raw);
    const int r = blockIdx.x;
    const int p0 = a.rb_row0[r], rows = a.rb_row0[r + 1] - p0;
    for (int i = threadIdx.x; i < rows; i += 512) yl[i] = 0.0;
    __syncthreads();
    const int s0 = a.rb_seg0[r], s1 = a.rb_seg0[r + 1];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    constexpr int LPS = kTpSeg / 8, SPW = 64 / LPS;
    const int sub = lane / LPS, off = (lane % LPS) * 8;
    const T *val = static_cast<const T *>(a.val), *xs = static_cast<const T *>(a.xs);
#pragma unroll 2
    for (int g = s0 + wave * SPW; g < s1; g += 8 * SPW) {
        const int seg = g + sub;
        if (seg < s1) {
            const size_t at = (size_t)seg * kTpSeg + off;
            const vec8 v = __builtin_nontemporal_load(reinterpret_cast<const vec8 *>(val + at));
            const tp_u16x8 lr = __builtin_nontemporal_load(reinterpret_cast<const tp_u16x8 *>(a.lrow + at));
            const vec8 xv = __builtin_nontemporal_load(reinterpret_cast<const vec8 *>(xs + at));
            // a tile's elements are in row order: consecutive elements of one lane that share a row are added up first (in f64, exactly) and cost ONE atomic.  The long
            // rows at the head of the sorted order then collide far less on one LDS word: ljournal-2008 0.2108 -> 0.1747 ms, the uniform-column variant unchanged (0.174)
            unsigned cur = lr[0];
            double run = (double)((float)v[0] * (float)xv[0]);
            bool one_run = true;                               // all 8 elements of this lane belong to one row (the inside of a hub row's run)
#pragma unroll
            for (int j = 1; j < 8; ++j) {
                const unsigned r = lr[j];
                const double p = (double)((float)v[j] * (float)xv[j]);
                if (r == cur) run += p;
                else {
                    if (cur != kTpPadRow) __hip_atomic_fetch_add(yl + cur, run, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                    cur = r; run = p; one_run = false;
                }
            }
            // ... and ACROSS the lanes of one segment (8 lanes = one tile's 64 consecutive elements, rows ascending, so equal rows are neighbours): lanes that each hold
            // one run of the same row -- the inside of a hub row's run -- are summed by a segmented scan (DPP row_shr 1, 2, 4) and the last lane of the run adds the
            // total: one atomic per run instead of one per lane.  (Not across segments: two tiles of one instruction may both hold the row, with other rows between.)
            const unsigned key = one_run && cur != kTpPadRow ? cur : 0x10000u + (unsigned)lane;
            const int ls = lane % LPS;
#define DASP_TP_SCAN(D, CTRL) if constexpr (D < LPS) { const unsigned k2 = (unsigned)__builtin_amdgcn_update_dpp((int)key, (int)key, CTRL, 0xf, 0xf, false); const double r2 = dpp_mov_f64<CTRL>(run); if (ls >= D && k2 == key) run += r2; }
            DASP_TP_SCAN(1, 0x111) DASP_TP_SCAN(2, 0x112) DASP_TP_SCAN(4, 0x114)
#undef DASP_TP_SCAN
            const unsigned knext = (unsigned)__builtin_amdgcn_update_dpp((int)key, (int)key, 0x101, 0xf, 0xf, false);       // row_shl:1 -- the next lane's key (same segment: active whenever this lane is)
            const bool last = ls == LPS - 1 || knext != key;
            if (cur != kTpPadRow && last) __hip_atomic_fetch_add(yl + cur, run, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        }
    }
    __syncthreads();
    if (acc) { for (int i = threadIdx.x; i < rows; i += 512) y[p0 + i] = (T)((float)y[p0 + i] + (float)yl[i]); }
    else for (int i = threadIdx.x; i < rows; i += 512) y[p0 + i] = (T)(float)yl[i];
}

// ------------------------------------------------------------------ column-blocked long rows of a column-panel plan (plan.hpp struct LongCB)
// One workgroup (1024 threads) per unit: the column block's slice of x -> LDS, then one wave per (row, block) piece: value x LDS-x, 16 bytes of values per lane
// and step, wave sum -> partial[piece].  Pads carry local column 0xFFFF and never touch x.
template <class T>
__global__ __launch_bounds__(1024) void dasp_lcb_kernel(LcbDev a, const T *__restrict__ x)
{
    extern __shared__ __attribute__((aligned(16))) char lds_raw[];
    using part_t = typename Tr<T>::part_t;
    constexpr int A = 16 / (int)sizeof(T);
    typedef T vecA __attribute__((ext_vector_type(A)));
    typedef unsigned short colA __attribute__((ext_vector_type(A)));
    T *xl = reinterpret_cast<T *>(lds_raw);
    const int u = blockIdx.x;
    const int c = a.unit[3 * u], q0 = a.unit[3 * u + 1], q1 = a.unit[3 * u + 2];
    const int c0 = c * a.cb, len = min(a.cb, a.xlen - c0);
    if ((reinterpret_cast<uintptr_t>(x) & 15) == 0) {
        for (int i = threadIdx.x * A; i < len; i += 1024 * A) {
            if (i + A <= len) *reinterpret_cast<vecA *>(xl + i) = *reinterpret_cast<const vecA *>(x + c0 + i);
            else for (int j = i; j < len; ++j) xl[j] = x[c0 + j];
        }
    } else for (int i = threadIdx.x; i < len; i += 1024) xl[i] = x[c0 + i];
    __syncthreads();
    // the unit's steps, wave-strided (f64: a wave per step of 128 elements; f16: a quarter wave per step, four steps per wave instruction): every step's loads are
    // independent of every other's -- no per-piece chain of latencies (one wave per piece ran at 2.4 TB/s however far its loop was unrolled) -- and its sum is
    // parked in LDS; afterwards one thread per piece adds the piece's step sums in order: no atomics, the same bits in every run
    constexpr int G = kLcbStep / A;                       // lanes per step: 64 (f64) / 16 (f16)
    constexpr int SPW = 64 / G;                           // steps per wave instruction
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int g = lane / G, l = lane % G;
    part_t *stepsum = reinterpret_cast<part_t *>(lds_raw + (((size_t)a.cb * sizeof(T) + 15) & ~size_t(15)));
    const int np = q1 - q0;
    const T *val = static_cast<const T *>(a.val);
    const int S0 = a.ptr[q0] / kLcbStep, S1 = a.ptr[q1] / kLcbStep;
    constexpr int U = 4;                                  // steps in flight per lane group: all their loads are issued before the first product
    for (int sb = S0 + wave * SPW * U; sb < S1; sb += 16 * SPW * U) {
        vecA v[U]; colA lc[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int st = min(sb + u * SPW + g, S1 - 1);          // (a step past the end re-reads the last one; its sum is dropped below)
            const size_t e = (size_t)st * kLcbStep + (size_t)l * A;
            v[u] = __builtin_nontemporal_load(reinterpret_cast<const vecA *>(val + e));
            lc[u] = __builtin_nontemporal_load(reinterpret_cast<const colA *>(a.lcol + e));
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            part_t s = 0;
#pragma unroll
            for (int j = 0; j < A; ++j) {
                const T xv = xl[lc[u][j] == kLcbPadCol ? 0 : lc[u][j]];
                s += lc[u][j] == kLcbPadCol ? (part_t)0 : (part_t)v[u][j] * (part_t)xv;
            }
            if constexpr (G == 64) s = wave_sum(s);
            else if constexpr (sizeof(part_t) == 8) { s += dpp_mov_f64<0x128>(s); s += dpp_mov_f64<0x124>(s); s += dpp_mov_f64<0x122>(s); s += dpp_mov_f64<0x121>(s); }
            else { s += dpp_mov_f32<0x128>(s); s += dpp_mov_f32<0x124>(s); s += dpp_mov_f32<0x122>(s); s += dpp_mov_f32<0x121>(s); }
            if (l == 0 && sb + u * SPW + g < S1) stepsum[sb + u * SPW + g - S0] = s;
        }
    }
    __syncthreads();
    part_t *partial = static_cast<part_t *>(a.partial);
    for (int i = threadIdx.x; i < np; i += 1024) {
        part_t s = 0;
        for (int st = a.ptr[q0 + i] / kLcbStep - S0, se = a.ptr[q0 + i + 1] / kLcbStep - S0; st < se; ++st) s += stepsum[st];
        partial[q0 + i] = s;
    }
}
// one wave per long row: its n_cb partial sums -> the row's slot of panel 0's partial buffer (no panel writes it: the row is empty there)
// acc (the two-phase hybrid, whose phase 2 has stored 0 -- or left y as it was in accumulate mode -- at the hub rows' positions): out[dst] += the row's sum
template <class T>
__global__ __launch_bounds__(256) void dasp_lcb_reduce_kernel(LcbDev a, T *__restrict__ part0, int acc)
{
    using part_t = typename Tr<T>::part_t;
    const int lane = threadIdx.x & 63;
    const int i = blockIdx.x * kWavesPerWG + (threadIdx.x >> 6);
    if (i >= a.n_rows) return;
    const part_t *partial = static_cast<const part_t *>(a.partial);
    part_t s = 0;
    for (int c = lane; c < a.n_cb; c += kWave) s += partial[(size_t)c * (size_t)a.n_rows + (size_t)i];
    s = wave_sum(s);
    if (lane == 0) { const int d = a.row_dst[i]; part0[d] = acc ? (T)((part_t)part0[d] + s) : (T)s; }
}

#ifdef DASP_STAMPS
int g_stamp_launch = -1;          // host: the number the next stamped launch carries (-1: stamps off)
}  // namespace dasp
// tools/stamp_probe.py: where the stamped kernels write (device pointer; rec = cap records of 12 words) and the first launch number
extern "C" int dasp_debug_set_stamps(void *rec, unsigned cap)
{
    dasp::StampBuf b{static_cast<unsigned long long *>(rec), cap};
    dasp::g_stamp_launch = rec ? 0 : -1;
    return hipMemcpyToSymbol(HIP_SYMBOL(dasp::g_stamps), &b, sizeof b) == hipSuccess ? 0 : -1;
}
namespace dasp {
#endif

// ------------------------------------------------------------------ MFMA lane-map self test
__global__ void selftest_f64_kernel(double *D)
{
    const int l = threadIdx.x;
    const double A = (double)((l & 15) * 4 + (l >> 4) + 1);        // A[i][k] = 4i + k + 1
    const double B = (double)(((l >> 4) + 1) * 100 + (l & 15));     // B[k][j] = 100(k+1) + j
    f64x4 acc = {0, 0, 0, 0};
    acc = __builtin_amdgcn_mfma_f64_16x16x4f64(A, B, acc, 0, 0, 0);
    for (int r = 0; r < 4; ++r) D[((l >> 4) + 4 * r) * 16 + (l & 15)] = acc[r];
}
__global__ void selftest_f16_kernel(float *D)
{
    const int l = threadIdx.x;
    f16x4 A, B;
    for (int j = 0; j < 4; ++j) {
        const int k = 4 * (l >> 4) + j;
        A[j] = (_Float16)(float)(((l & 15) + 2 * k) % 7 + 1);       // A[i][k]
        B[j] = (_Float16)(float)((3 * k + (l & 15)) % 5 + 1);       // B[k][j]
    }
    f32x4 acc = {0, 0, 0, 0};
    acc = __builtin_amdgcn_mfma_f32_16x16x16f16(A, B, acc, 0, 0, 0);
    for (int r = 0; r < 4; ++r) D[(4 * (l >> 4) + r) * 16 + (l & 15)] = acc[r];
}

// ------------------------------------------------------------------ host side

#define HIP_TRY(expr)                                                                          \
    do {                                                                                       \
        hipError_t e_ = (expr);                                                                \
        if (e_ != hipSuccess) {                                                                \
            set_error(std::string(#expr) + ": " + hipGetErrorString(e_));                      \
            return e_ == hipErrorNoDevice ? DASP_ERR_NO_DEVICE : DASP_ERR_HIP;                 \
        }                                                                                      \
    } while (0)


// ---- what upload.cpp needs to know about the kernels (it is host code and never names a kernel itself)
// windowed plans with more than the default 64 KiB of dynamic LDS: the limit must be raised per kernel.  Done at upload (for both
// cache-policy variants), not in the launch path, so that dasp_plan_spmv stays free of anything a stream capture would reject; the
// attribute belongs to the kernel, not to the plan: always the device maximum, or a later plan with narrower windows would lower
// the limit under an earlier one with wider windows
int spmv_kernel_allow_full_lds(int precision, bool c16)
{
    const int bytes = kWinLdsMax;          // + the window kernels' static LDS (the unit counter) = under the 160 KiB of a CU
    hipError_t e1, e2;
#define DASP_ATTR(TT, NTV, CV) hipFuncSetAttribute(reinterpret_cast<const void *>(&dasp_spmv_kernel<TT, NTV, CV, true>), hipFuncAttributeMaxDynamicSharedMemorySize, bytes)
    if (precision == 64) { e1 = c16 ? DASP_ATTR(double, true, true) : DASP_ATTR(double, true, false); e2 = c16 ? DASP_ATTR(double, false, true) : DASP_ATTR(double, false, false); }
    else { e1 = c16 ? DASP_ATTR(_Float16, true, true) : DASP_ATTR(_Float16, true, false); e2 = c16 ? DASP_ATTR(_Float16, false, true) : DASP_ATTR(_Float16, false, false); }
#undef DASP_ATTR
    HIP_TRY(e1);
    HIP_TRY(e2);
    if (precision == 64) HIP_TRY(c16 ? hipFuncSetAttribute(reinterpret_cast<const void *>(&dasp_spmv_win1_kernel<double, true>), hipFuncAttributeMaxDynamicSharedMemorySize, bytes)
                                     : hipFuncSetAttribute(reinterpret_cast<const void *>(&dasp_spmv_win1_kernel<double, false>), hipFuncAttributeMaxDynamicSharedMemorySize, bytes));
    else HIP_TRY(c16 ? hipFuncSetAttribute(reinterpret_cast<const void *>(&dasp_spmv_win1_kernel<_Float16, true>), hipFuncAttributeMaxDynamicSharedMemorySize, bytes)
                     : hipFuncSetAttribute(reinterpret_cast<const void *>(&dasp_spmv_win1_kernel<_Float16, false>), hipFuncAttributeMaxDynamicSharedMemorySize, bytes));
    return DASP_OK;
}
// workgroups of the non-windowed f16 kernel one CU holds at a time (0: unknown)
int spmv_kernel_f16_resident(bool c16)
{
    const void *fn = c16 ? reinterpret_cast<const void *>(&dasp_spmv_kernel<_Float16, true, true, false>)
                         : reinterpret_cast<const void *>(&dasp_spmv_kernel<_Float16, true, false, false>);
    int fit = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&fit, fn, kWave * kWavesPerWG, 0) != hipSuccess) fit = 0;
    (void)hipGetLastError();
    return fit;
}

template <class T>
static int launch_typed(Plan &p, const DevArgs &a, hipStream_t s)
{
    const int grid = a.wg_long + a.wg_med + a.wg_short + a.wg_rt;
    const bool nt = p.dev->nt;
    if (int rc = sync_dev_args(p)) return rc;          // (a memcmp: the device copy follows DevicePlan::args)
#ifdef DASP_STAMPS
    const CallArgs c{static_cast<const DevArgs *>(p.dev->dargs), a.x, a.y, a.acc, a.ywt, g_stamp_launch >= 0 ? g_stamp_launch++ : -1};
#else
    const CallArgs c{static_cast<const DevArgs *>(p.dev->dargs), a.x, a.y, a.acc, a.ywt};
#endif
    if (a.wg_rt > 0) {      // a column panel with row tiles (never windowed, never with one-byte ids: plan.cpp build_panels)
        const size_t lds = (size_t)kWavesPerWG * kRowTile * (size_t)a.rt_max * sizeof(typename Tr<T>::part_t);
        if (nt && p.cid16) hipLaunchKernelGGL((dasp_spmv_rt_kernel<T, true, true>), dim3(grid), dim3(256), lds, s, c);
        else if (nt) hipLaunchKernelGGL((dasp_spmv_rt_kernel<T, true, false>), dim3(grid), dim3(256), lds, s, c);
        else if (p.cid16) hipLaunchKernelGGL((dasp_spmv_rt_kernel<T, false, true>), dim3(grid), dim3(256), lds, s, c);
        else hipLaunchKernelGGL((dasp_spmv_rt_kernel<T, false, false>), dim3(grid), dim3(256), lds, s, c);
    } else if (grid > 0) {
        const size_t lds = p.windowed ? (size_t)p.lds_bytes : 0;
        const bool c16 = p.cid16;
#define DASP_FOR_EACH(M) \
        if (nt && c16 && p.windowed) { M(true, true, true); } else if (nt && c16) { M(true, true, false); } \
        else if (nt && p.windowed) { M(true, false, true); } else if (nt) { M(true, false, false); } \
        else if (c16 && p.windowed) { M(false, true, true); } else if (c16) { M(false, true, false); } \
        else if (p.windowed) { M(false, false, true); } else { M(false, false, false); }
#define DASP_LAUNCH(NTV, CV, WINV) hipLaunchKernelGGL((dasp_spmv_kernel<T, NTV, CV, WINV>), dim3(grid), dim3(kWave * a.wpw), lds, s, c)
        if (p.windowed && p.dev->win1 && !nt) {                             // at most one window workgroup per CU: the 128-register build
            if (c16) hipLaunchKernelGGL((dasp_spmv_win1_kernel<T, true>), dim3(grid), dim3(kWave * a.wpw), lds, s, c);
            else hipLaunchKernelGGL((dasp_spmv_win1_kernel<T, false>), dim3(grid), dim3(kWave * a.wpw), lds, s, c);
        } else if (sizeof(T) == 8 && c16 && !p.windowed && p.cnt_reg8 > 0) {      // plans with one-byte ids: their own instantiation
            if (p.dev->seven_waves) {
                if (nt) hipLaunchKernelGGL((dasp_spmv_kernel<double, true, true, false, true, 7>), dim3(grid), dim3(kWave * a.wpw), lds, s, c);
                else hipLaunchKernelGGL((dasp_spmv_kernel<double, false, true, false, true, 7>), dim3(grid), dim3(kWave * a.wpw), lds, s, c);
            }
            else if (nt) hipLaunchKernelGGL((dasp_spmv_kernel<double, true, true, false, true>), dim3(grid), dim3(kWave * a.wpw), lds, s, c);
            else hipLaunchKernelGGL((dasp_spmv_kernel<double, false, true, false, true>), dim3(grid), dim3(kWave * a.wpw), lds, s, c);
        } else if (!p.windowed && p.dev->long16 && !(sizeof(T) == 8 && p.dev->seven_waves)) {      // narrow long pieces that matter: the builds that read their 16-bit ids
            if (nt && c16) hipLaunchKernelGGL((dasp_spmv_kernel<T, true, true, false, false, 0, true>), dim3(grid), dim3(kWave * a.wpw), lds, s, c);
            else if (nt) hipLaunchKernelGGL((dasp_spmv_kernel<T, true, false, false, false, 0, true>), dim3(grid), dim3(kWave * a.wpw), lds, s, c);
            else if (c16) hipLaunchKernelGGL((dasp_spmv_kernel<T, false, true, false, false, 0, true>), dim3(grid), dim3(kWave * a.wpw), lds, s, c);
            else hipLaunchKernelGGL((dasp_spmv_kernel<T, false, false, false, false, 0, true>), dim3(grid), dim3(kWave * a.wpw), lds, s, c);
        } else if (sizeof(T) == 8 && !p.windowed && p.dev->seven_waves) {
            if (nt && c16) hipLaunchKernelGGL((dasp_spmv_kernel<double, true, true, false, false, 7>), dim3(grid), dim3(kWave * a.wpw), lds, s, c);
            else if (nt) hipLaunchKernelGGL((dasp_spmv_kernel<double, true, false, false, false, 7>), dim3(grid), dim3(kWave * a.wpw), lds, s, c);
            else if (c16) hipLaunchKernelGGL((dasp_spmv_kernel<double, false, true, false, false, 7>), dim3(grid), dim3(kWave * a.wpw), lds, s, c);
            else hipLaunchKernelGGL((dasp_spmv_kernel<double, false, false, false, false, 7>), dim3(grid), dim3(kWave * a.wpw), lds, s, c);
        } else
        DASP_FOR_EACH(DASP_LAUNCH)
#undef DASP_LAUNCH
#undef DASP_FOR_EACH
    }
    if (a.n_multi > 0)
        hipLaunchKernelGGL((dasp_long_reduce_kernel<T>), dim3(a.n_multi), dim3(256), 0, s, c);
    HIP_TRY(hipGetLastError());
    return DASP_OK;
}

int set_stream_policy(Plan &p, int policy)
{
    if (policy < 0 || policy > 2) { set_error("stream_policy must be 0, 1 or 2"); return DASP_ERR_ARG; }
    p.opt.stream_policy = policy;
    if (p.dev) p.dev->nt = policy == 2 || (policy != 1 && p.stats.data_X > kStreamBytes);
    // panels follow the whole matrix: auto means non-temporal when the sum of the panels streams from HBM
    const int sub = policy != 0 ? policy : (p.stats.data_X > kStreamBytes ? 2 : 1);
    for (auto &h : p.panels) if (int rc = set_stream_policy(h->impl, sub)) return rc;
    return DASP_OK;
}

int tp_kernels_allow_lds()
{
    HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void *>(&dasp_tp_expand_kernel<_Float16>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void *>(&dasp_tp_reduce_kernel<_Float16>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void *>(&dasp_lcb_kernel<_Float16>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void *>(&dasp_lcb_kernel<double>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    return DASP_OK;
}

// the panels of a column-panel plan in one launch (+ one stage-2 launch when some panel cut a long row).  DASP_OK, an error, or 1 when the panels
// cannot share one instantiation of the kernel (windows, one-byte ids, mixed id widths, more than kMaxMergedPanels) or DASP_PANELS_MERGED=0 asks for the old form
template <class T>
static int launch_panels_merged_typed(Plan &p, const void *dX, char *part, size_t stride_bytes, hipStream_t s)
{
    const int np = (int)p.panels.size();
    PanelCall c{}, r{};
    c.x = dX; c.part = part; c.stride_bytes = stride_bytes; c.np = np;
    int grid = 0, grid2 = 0, rt_max = 0;
    const Plan &p0 = p.panels[0]->impl;
    for (int k = 0; k < np; ++k) {
        Plan &q = p.panels[(size_t)k]->impl;
        if (!q.dev || !q.dev->arena || q.windowed || q.two_phase || !q.panels.empty() || q.cnt_reg8 > 0 || q.cid16 != p0.cid16 || q.dev->nt != p0.dev->nt || q.dev->args.med_stride != p0.dev->args.med_stride) return 1;
        if (int rc = sync_dev_args(q)) return rc;
        const DevArgs &a = q.dev->args;
        grid += a.wg_long + a.wg_med + a.wg_short + a.wg_rt;
        grid2 += (a.n_multi + kWavesPerWG - 1) / kWavesPerWG;
        rt_max = std::max(rt_max, a.wg_rt > 0 ? a.rt_max : 0);
        c.plan[k] = static_cast<const DevArgs *>(q.dev->dargs); c.wg_end[k] = grid;
        r.plan[k] = c.plan[k]; r.wg_end[k] = grid2;
    }
    r.x = dX; r.part = part; r.stride_bytes = stride_bytes; r.np = np;
    const size_t lds = (size_t)kWavesPerWG * kRowTile * (size_t)rt_max * sizeof(typename Tr<T>::part_t);
    const bool nt = p0.dev->nt, c16 = p0.cid16;
    if (grid > 0) {
        if (nt && c16) hipLaunchKernelGGL((dasp_spmv_panels_kernel<T, true, true>), dim3(grid), dim3(256), lds, s, c);
        else if (nt) hipLaunchKernelGGL((dasp_spmv_panels_kernel<T, true, false>), dim3(grid), dim3(256), lds, s, c);
        else if (c16) hipLaunchKernelGGL((dasp_spmv_panels_kernel<T, false, true>), dim3(grid), dim3(256), lds, s, c);
        else hipLaunchKernelGGL((dasp_spmv_panels_kernel<T, false, false>), dim3(grid), dim3(256), lds, s, c);
    }
    if (grid2 > 0) hipLaunchKernelGGL((dasp_long_reduce_panels_kernel<T>), dim3(grid2), dim3(256), 0, s, r);
    HIP_TRY(hipGetLastError());
    return DASP_OK;
}
static int launch_panels_merged(Plan &p, const void *dX, char *part, size_t stride_bytes, hipStream_t s)
{
    static const bool off = [] { const char *e = std::getenv("DASP_PANELS_MERGED"); return e && std::atoi(e) == 0; }();      // A/B knob
    if (off || p.panels.empty() || p.panels.size() > (size_t)kMaxMergedPanels) return 1;
    return p.precision == 64 ? launch_panels_merged_typed<double>(p, dX, part, stride_bytes, s) : launch_panels_merged_typed<_Float16>(p, dX, part, stride_bytes, s);
}

int launch_spmv(Plan &p, const void *dX, void *dY, void *stream, bool accumulate)
{
    if (!p.dev || !p.dev->arena) { set_error("plan not uploaded"); return DASP_ERR_STATE; }
    if (!dX || !dY) { set_error("null device pointer"); return DASP_ERR_ARG; }
    if (p.two_phase) {
        const TpDev &a = p.dev->tp;
        hipStream_t s = static_cast<hipStream_t>(stream);
        // the hybrid's hub rows (Plan::lcb): their streaming kernel, then -- BEHIND phase 2, which stores 0 at their positions or leaves y alone in accumulate mode -- their
        // per-row sums added into y
        const bool hub = p.lcb.n_rows() > 0;
        const LcbDev &q = p.dev->lcb;
        if (hub)
            hipLaunchKernelGGL((dasp_lcb_kernel<_Float16>), dim3(q.n_units), dim3(1024), (size_t)q.cb * 2 + 16 + (size_t)(kLcbUnitElems / kLcbStep + kLcbUnitPieces) * 8, s, q, static_cast<const _Float16 *>(dX));
        if (a.n_units > 0)
            hipLaunchKernelGGL((dasp_tp_expand_kernel<_Float16>), dim3(a.n_units), dim3(512), (size_t)a.cb * 2, s, a, static_cast<const _Float16 *>(dX));
        if (a.n_rb > 0)
            hipLaunchKernelGGL((dasp_tp_reduce_kernel<_Float16>), dim3(a.n_rb), dim3(512), (size_t)a.rb_max * 8, s, a, static_cast<_Float16 *>(dY), accumulate ? 1 : 0);
        if (hub) {
            hipLaunchKernelGGL((dasp_lcb_reduce_kernel<_Float16>), dim3((q.n_rows + kWavesPerWG - 1) / kWavesPerWG), dim3(256), 0, s, q, static_cast<_Float16 *>(dY), 1);
        }
        HIP_TRY(hipGetLastError());
        return DASP_OK;
    }
    if (!p.panels.empty()) {
        const size_t vb = (size_t)p.geo.vbytes, stride = p.dev->ypart_stride;
        char *part = static_cast<char *>(p.dev->arena);
        hipStream_t s = static_cast<hipStream_t>(stream);
        // the hub rows (Plan::lcb): column blocks of x staged in LDS, their result into panel 0's slots of the partial buffer -- BEHIND the panels' launch on the same
        // stream: a panel's row tiles store 0 at the positions of rows that are empty in it, the hub rows' too.  (r6, measured and not kept: the two hub kernels on a
        // stream of the plan's own beside the panels, fork / join by events -- powerlaw_1M f64 375 -> 380 us in back-to-back launches, 369 when captured in a graph;
        // both kernels are bound by what ONE CU keeps in flight, so CUs given to one are taken from the other: profiles/r06_hub_rows.md)
        auto launch_hub = [&]() {
            const LcbDev &q = p.dev->lcb;
            if (p.precision == 64) {
                hipLaunchKernelGGL((dasp_lcb_kernel<double>), dim3(q.n_units), dim3(1024), (size_t)q.cb * 8 + 16 + (size_t)(kLcbUnitElems / kLcbStep + kLcbUnitPieces) * 8, s, q, static_cast<const double *>(dX));
                hipLaunchKernelGGL((dasp_lcb_reduce_kernel<double>), dim3((q.n_rows + kWavesPerWG - 1) / kWavesPerWG), dim3(256), 0, s, q, reinterpret_cast<double *>(part), 0);
            } else {
                hipLaunchKernelGGL((dasp_lcb_kernel<_Float16>), dim3(q.n_units), dim3(1024), (size_t)q.cb * 2 + 16 + (size_t)(kLcbUnitElems / kLcbStep + kLcbUnitPieces) * 8, s, q, static_cast<const _Float16 *>(dX));
                hipLaunchKernelGGL((dasp_lcb_reduce_kernel<_Float16>), dim3((q.n_rows + kWavesPerWG - 1) / kWavesPerWG), dim3(256), 0, s, q, reinterpret_cast<_Float16 *>(part), 0);
            }
        };
        if (int rc = launch_panels_merged(p, dX, part, stride * vb, s)) {
            if (rc != 1) return rc;          // 1: the panels do not share one kernel instantiation -- one launch (+ stage 2) per panel, as before r5
            for (size_t k = 0; k < p.panels.size(); ++k)
                if (int rc2 = launch_spmv(p.panels[k]->impl, dX, part + k * stride * vb, stream, false)) return rc2;
        }
        if (p.lcb.n_rows() > 0) launch_hub();
        const int np = (int)p.panels.size(), m = p.m;
        const bool wide = (reinterpret_cast<uintptr_t>(dY) & 15) == 0;
        if (m > 0) {
            if (p.precision == 64) {
                if (wide) hipLaunchKernelGGL((dasp_panel_sum_kernel<double, 2>), dim3((m + 511) / 512), dim3(256), 0, s, (const double *)part, stride, np, (double *)dY, m, accumulate ? 1 : 0);
                else hipLaunchKernelGGL((dasp_panel_sum_kernel<double, 1>), dim3((m + 255) / 256), dim3(256), 0, s, (const double *)part, stride, np, (double *)dY, m, accumulate ? 1 : 0);
            } else {
                if (wide) hipLaunchKernelGGL((dasp_panel_sum_kernel<_Float16, 8>), dim3((m + 2047) / 2048), dim3(256), 0, s, (const _Float16 *)part, stride, np, (_Float16 *)dY, m, accumulate ? 1 : 0);
                else hipLaunchKernelGGL((dasp_panel_sum_kernel<_Float16, 1>), dim3((m + 255) / 256), dim3(256), 0, s, (const _Float16 *)part, stride, np, (_Float16 *)dY, m, accumulate ? 1 : 0);
            }
        }
        HIP_TRY(hipGetLastError());
        return DASP_OK;
    }
    if (p.windowed && (reinterpret_cast<uintptr_t>(dX) & 15)) {   // the window copy uses 16-byte loads from x + cmin (cmin is 16-byte granular)
        set_error("dX must be 16-byte aligned for a plan with LDS-staged x windows"); return DASP_ERR_ARG;
    }
    DevArgs a = p.dev->args;
    a.x = dX; a.y = dY; a.acc = accumulate ? 1 : 0;
    // one block / tile / piece per wave and then the wave ends: f64 plans without x windows and without a striding medium range
    a.ywt = p.dev->nt && p.precision == 64 && !p.windowed && !a.med_stride && !p.panel ? 1 : 0;
    if (const char *e = std::getenv("DASP_Y_WT")) a.ywt = a.ywt && std::atoi(e) != 0;      // A/B knob
    hipStream_t s = static_cast<hipStream_t>(stream);
    return p.precision == 64 ? launch_typed<double>(p, a, s) : launch_typed<_Float16>(p, a, s);
}

namespace {
// event pair / capture objects released on every return path
struct EventPair {
    hipEvent_t e0 = nullptr, e1 = nullptr;
    ~EventPair() { if (e0) (void)hipEventDestroy(e0); if (e1) (void)hipEventDestroy(e1); }
};
struct GraphHolder {
    hipGraph_t graph = nullptr; hipGraphExec_t exec = nullptr; hipStream_t own = nullptr;
    ~GraphHolder()
    {
        if (exec) (void)hipGraphExecDestroy(exec);
        if (graph) (void)hipGraphDestroy(graph);
        if (own) (void)hipStreamDestroy(own);
    }
};
}  // namespace

int time_spmv(Plan &p, const void *dX, void *dY, void *stream, int warmup, int iters, double *wall_ms, double *event_ms)
{
    hipStream_t s = static_cast<hipStream_t>(stream);
    for (int i = 0; i < warmup; ++i) if (int rc = launch_spmv(p, dX, dY, stream, false)) return rc;
    HIP_TRY(hipStreamSynchronize(s));
    EventPair ev;
    HIP_TRY(hipEventCreate(&ev.e0));
    HIP_TRY(hipEventCreate(&ev.e1));
    const auto t0 = std::chrono::steady_clock::now();
    HIP_TRY(hipEventRecord(ev.e0, s));
    for (int i = 0; i < iters; ++i) if (int rc = launch_spmv(p, dX, dY, stream, false)) return rc;
    HIP_TRY(hipEventRecord(ev.e1, s));
    HIP_TRY(hipStreamSynchronize(s));
    const auto t1 = std::chrono::steady_clock::now();
    float ms = 0;
    HIP_TRY(hipEventElapsedTime(&ms, ev.e0, ev.e1));
    if (iters > 0) {
        if (wall_ms) *wall_ms = std::chrono::duration<double, std::milli>(t1 - t0).count() / iters;
        if (event_ms) *event_ms = (double)ms / iters;
    }
    return DASP_OK;
}

int time_spmv_each(Plan &p, const void *dX, void *dY, void *stream, int warmup, int iters, float *ms_each)
{
    if (iters <= 0 || !ms_each) { set_error("time_each: iters > 0 and an output array"); return DASP_ERR_ARG; }
    hipStream_t s = static_cast<hipStream_t>(stream);
    for (int i = 0; i < warmup; ++i) if (int rc = launch_spmv(p, dX, dY, stream, false)) return rc;
    HIP_TRY(hipStreamSynchronize(s));
    struct Events {
        std::vector<hipEvent_t> e;
        ~Events() { for (hipEvent_t x : e) if (x) (void)hipEventDestroy(x); }
    } ev;
    ev.e.assign((size_t)iters + 1, nullptr);
    for (hipEvent_t &x : ev.e) HIP_TRY(hipEventCreate(&x));
    HIP_TRY(hipEventRecord(ev.e[0], s));
    for (int i = 0; i < iters; ++i) {
        if (int rc = launch_spmv(p, dX, dY, stream, false)) return rc;
        HIP_TRY(hipEventRecord(ev.e[(size_t)i + 1], s));
    }
    HIP_TRY(hipStreamSynchronize(s));
    for (int i = 0; i < iters; ++i) HIP_TRY(hipEventElapsedTime(&ms_each[i], ev.e[(size_t)i], ev.e[(size_t)i + 1]));
    return DASP_OK;
}

// same protocol with the launches captured once into a hipGraph of `batch` SpMVs and replayed: removes the
// per-launch host cost (3-4 us) that bounds back-to-back launches of small matrices; the kernels are unchanged.
int time_spmv_graph(Plan &p, const void *dX, void *dY, void *stream, int warmup, int iters, int batch, double *wall_ms, double *event_ms)
{
    if (batch <= 0) batch = 1;
    hipStream_t cap = static_cast<hipStream_t>(stream);
    GraphHolder g;
    if (cap == nullptr) {                       // the legacy null stream cannot be captured
        HIP_TRY(hipStreamCreateWithFlags(&g.own, hipStreamNonBlocking));
        cap = g.own;
    }
    HIP_TRY(hipStreamBeginCapture(cap, hipStreamCaptureModeThreadLocal));
    int rc = DASP_OK;
    for (int i = 0; i < batch && rc == DASP_OK; ++i) rc = launch_spmv(p, dX, dY, cap, false);
    const hipError_t ee = hipStreamEndCapture(cap, &g.graph);      // always end the capture, even after a failed launch
    if (rc != DASP_OK) return rc;
    HIP_TRY(ee);
    HIP_TRY(hipGraphInstantiate(&g.exec, g.graph, nullptr, nullptr, 0));
    const int reps = (iters + batch - 1) / batch, wreps = (warmup + batch - 1) / batch;
    for (int i = 0; i < wreps; ++i) HIP_TRY(hipGraphLaunch(g.exec, cap));
    HIP_TRY(hipStreamSynchronize(cap));
    EventPair ev;
    HIP_TRY(hipEventCreate(&ev.e0));
    HIP_TRY(hipEventCreate(&ev.e1));
    const auto t0 = std::chrono::steady_clock::now();
    HIP_TRY(hipEventRecord(ev.e0, cap));
    for (int i = 0; i < reps; ++i) HIP_TRY(hipGraphLaunch(g.exec, cap));
    HIP_TRY(hipEventRecord(ev.e1, cap));
    HIP_TRY(hipStreamSynchronize(cap));
    const auto t1 = std::chrono::steady_clock::now();
    float ms = 0;
    HIP_TRY(hipEventElapsedTime(&ms, ev.e0, ev.e1));
    const double n = (double)reps * batch;
    if (wall_ms) *wall_ms = std::chrono::duration<double, std::milli>(t1 - t0).count() / n;
    if (event_ms) *event_ms = (double)ms / n;
    return DASP_OK;
}

int selftest_mfma()
{
    if (int rc = require_device()) return rc;
    double *dD = nullptr; float *dF = nullptr;
    HIP_TRY(hipMalloc(&dD, 256 * sizeof(double)));
    HIP_TRY(hipMalloc(&dF, 256 * sizeof(float)));
    hipLaunchKernelGGL(selftest_f64_kernel, dim3(1), dim3(64), 0, 0, dD);
    hipLaunchKernelGGL(selftest_f16_kernel, dim3(1), dim3(64), 0, 0, dF);
    HIP_TRY(hipGetLastError());
    double hD[256]; float hF[256];
    HIP_TRY(hipMemcpy(hD, dD, sizeof hD, hipMemcpyDeviceToHost));
    HIP_TRY(hipMemcpy(hF, dF, sizeof hF, hipMemcpyDeviceToHost));
    (void)hipFree(dD); (void)hipFree(dF);
    int bad = 0;
    for (int i = 0; i < 16; ++i)
        for (int j = 0; j < 16; ++j) {
            double e = 0;
            for (int k = 0; k < 4; ++k) e += (double)(4 * i + k + 1) * (double)(100 * (k + 1) + j);
            if (hD[i * 16 + j] != e) bad |= 1;
            float f = 0;
            for (int k = 0; k < 16; ++k) f += (float)((i + 2 * k) % 7 + 1) * (float)((3 * k + j) % 5 + 1);
            if (hF[i * 16 + j] != f) bad |= 2;
        }
    if (bad) { set_error("MFMA lane map mismatch: mask " + std::to_string(bad)); return DASP_ERR_HIP; }
    return DASP_OK;
}

}  // namespace dasp

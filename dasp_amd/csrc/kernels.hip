// kernels.hip -- gfx950 (CDNA4, wave64) SpMV kernels for the DASP plan + device plumbing.
//
// One fused launch covers the three row categories by workgroup-index range, as the
// reference's dasp_spmv2 does (src/dasp_f64.h:77-484, src/dasp_f16.h:133-590):
//   [ long pieces | medium blocks | short tiles ]      256 threads = 4 waves, one unit per wave
// followed by long_reduce only when some long row was cut into several pieces
// (the reference's longPart_sum, src/dasp_f64.h:53-75).
//
// Medium / long rows use the DASP diagonal trick on the CDNA4 matrix cores: a chunk of
// 16 rows x K columns is fed as A = values, B = x[column ids] with the same element index on
// both operands, so D[i][i] accumulates row i's dot product (reference: m8n8k4 PTX MMA,
// src/utils.h:102-115; here v_mfma_f64_16x16x4_f64 / v_mfma_f32_16x16x16_f16).
// Lane maps (pinned on the device by dasp_selftest_mfma):
//   f64 16x16x4 : lane l holds A[l&15][l>>4], B[l>>4][l&15]; D reg r = D[(l>>4)+4r][l&15]
//   f16 16x16x16: lane l holds A[l&15][4(l>>4)+j], B[4(l>>4)+j][l&15], j<4; D reg r = D[4(l>>4)+r][l&15]
// Short rows (1..4 nonzeros) are uniform-length slabs: a lane owns whole rows, so the
// segmented dot product needs no cross-lane step; cross-lane sums (long rows, stage 2)
// use DPP row rotations + readlane.
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

namespace dasp {

typedef double f64x4 __attribute__((ext_vector_type(4)));
typedef double f64x2 __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
typedef int i32x4 __attribute__((ext_vector_type(4)));
typedef int i32x2 __attribute__((ext_vector_type(2)));
typedef unsigned short u16x4 __attribute__((ext_vector_type(4)));
typedef unsigned short u16x8 __attribute__((ext_vector_type(8)));
typedef const __attribute__((address_space(1))) unsigned char *gbyte_p;      // explicitly global: a select of two flat pointers is not inferred
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

// ------------------------------------------------------------------ device helpers

template <bool NT, class U>
__device__ __forceinline__ U ldg(const U *p)
{
    if constexpr (NT) return __builtin_nontemporal_load(p);
    else return *p;
}

template <int CTRL>
__device__ __forceinline__ double dpp_mov_f64(double v)
{
    int lo = __double2loint(v), hi = __double2hiint(v);
    lo = __builtin_amdgcn_update_dpp(lo, lo, CTRL, 0xf, 0xf, false);
    hi = __builtin_amdgcn_update_dpp(hi, hi, CTRL, 0xf, 0xf, false);
    return __hiloint2double(hi, lo);
}
template <int CTRL>
__device__ __forceinline__ float dpp_mov_f32(float v)
{
    return __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(v), __float_as_int(v), CTRL, 0xf, 0xf, false));
}
__device__ __forceinline__ double readlane_f64(double v, int l)
{
    int lo = __builtin_amdgcn_readlane(__double2loint(v), l);
    int hi = __builtin_amdgcn_readlane(__double2hiint(v), l);
    return __hiloint2double(hi, lo);
}
// sum over the 64 lanes of a wave, result uniform.  DPP row_ror:8,4,2,1 (0x120+n) make every lane of
// a 16-lane row hold that row's sum; the four rows are combined on the scalar side.
__device__ __forceinline__ double wave_sum(double v)
{
    v += dpp_mov_f64<0x128>(v);
    v += dpp_mov_f64<0x124>(v);
    v += dpp_mov_f64<0x122>(v);
    v += dpp_mov_f64<0x121>(v);
    return (readlane_f64(v, 0) + readlane_f64(v, 16)) + (readlane_f64(v, 32) + readlane_f64(v, 48));
}
__device__ __forceinline__ float wave_sum(float v)
{
    v += dpp_mov_f32<0x128>(v);
    v += dpp_mov_f32<0x124>(v);
    v += dpp_mov_f32<0x122>(v);
    v += dpp_mov_f32<0x121>(v);
    float a = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 0));
    float b = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 16));
    float c = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 32));
    float d = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 48));
    return (a + b) + (c + d);
}

// read-only plan tables indexed by wave-uniform values (block / piece / chunk numbers).  K = true (the fused multi-GPU step): read
// through the CONSTANT address space, so that they stay scalar loads -- that kernel waits on flags in memory, and with an atomic
// load or a fence anywhere in the function the compiler no longer proves global memory unclobbered and turns every such read into a
// vector load (the own-column product of an 8-way HV15R slice: 61 -> 92 us).  The tables are never written while a plan exists.
template <bool K, class U>
__device__ __forceinline__ U tab(const U *p, int i)
{
    if constexpr (K) return reinterpret_cast<const __attribute__((address_space(4))) U *>(reinterpret_cast<uintptr_t>(p))[i];
    else return p[i];
}

__device__ __forceinline__ int slot_of(const SlotMap &m, int t)
{
    const int p = t < m.split ? 0 : 1;
    const int u = p ? t - m.split : t;
    const int g = m.grp[p];
    return g ? m.base[p] + (u / g) * 2 * g + m.off[p] + u % g : m.base[p] + u;
}

// BATCH = chunks per software-pipeline batch, SHOT = longest unit issued in one shot (defaults chosen on the HBM-bound
// stand-ins, DESIGN.md 4.4: f64 rows of <= 32 nonzeros in one shot: cop20k_A 11.5 -> 10.9 us, HBM-bound stand-ins +0.5-1 %)
template <class T> struct Tr;
template <> struct Tr<double> {
    using acc_t = f64x4; using part_t = double;
    static constexpr int CHUNK = 64, SHORT_ROWS = 128, BATCH = kMedBatch64, SHOT = kMedShot64;
};
template <> struct Tr<_Float16> {
    using acc_t = f32x4; using part_t = float;
    static constexpr int CHUNK = 256, SHORT_ROWS = 256, BATCH = kMedBatch16, SHOT = kMedShot16;
};
static_assert(Tr<double>::BATCH == kMedBatch64 && Tr<double>::SHOT == kMedShot64 && Tr<_Float16>::BATCH == kMedBatch16 &&
              Tr<_Float16>::SHOT == kMedShot16 && kMedBatch64 % 2 == 0 && kMedBatch16 % 2 == 0, "the packers' pairing rule (plan.hpp) follows the kernel's batches");

// ---- chunk = one MFMA worth of elements in lane-linear order.  Loads, gathers and MFMAs are kept
// as separate branch-free stages so that a batch of chunks has all its streaming loads, then all
// its x gathers, in flight together.  Padded slots carry column id -1: the gather address is
// clamped to x[0] (always readable) and the gathered value replaced by 0, so a pad contributes an
// exact 0 whatever x holds (the reference multiplies 0 by x[0]: dasp_f64.h:1127-1128).
template <class T> struct Frag;
template <> struct Frag<double> { double a; int c; double b; };
template <> struct Frag<_Float16> { f16x4 a; i32x4 c; f16x4 b; };

// `at` = this lane's first element (f64: one element, f16: four consecutive ones)
template <bool NT>
__device__ __forceinline__ void frag_load_at(Frag<double> &f, const double *val, const int *cid, size_t at)
{
    f.a = ldg<NT>(val + at);
    f.c = ldg<NT>(cid + at);
}
template <bool NT>
__device__ __forceinline__ void frag_load_at(Frag<_Float16> &f, const _Float16 *val, const int *cid, size_t at)
{
    f.a = ldg<NT>(reinterpret_cast<const f16x4 *>(val + at));
    f.c = ldg<NT>(reinterpret_cast<const i32x4 *>(cid + at));
}
template <bool NT, class T>
__device__ __forceinline__ void frag_load(Frag<T> &f, const T *val, const int *cid, size_t e, int lane)
{
    frag_load_at<NT>(f, val, cid, e + (size_t)(Tr<T>::CHUNK / kWave) * lane);
}
// where x values come from: global memory, or the workgroup's window of x staged in LDS
template <class T>
struct XGlobal {
    const T *x;
    __device__ __forceinline__ T at(int c) const { return x[c < 0 ? 0 : c]; }       // pads read x[0], dropped below
};
template <class T>
struct XLds {
    const T *xw; int cmin;
    __device__ __forceinline__ T at(int c) const { return xw[c < 0 ? 0 : c - cmin]; }
};
// hybrid window: the densest span of the window's columns is in LDS, everything else is gathered from global memory.
// The two loads sit in divergent branches on purpose: a lane whose column is staged issues no global load.
template <class T>
struct XHyb {
    const T *xw; const T *xg; int cmin; unsigned len;
    __device__ __forceinline__ T at(int c) const
    {
        const unsigned o = (unsigned)(c - cmin);
        T v;
        if (c < 0) v = (T)0;                    // pad: dropped by the caller
        else if (o < len) v = xw[o];
        else v = xg[c];
        return v;
    }
};
template <class XV>
__device__ __forceinline__ void frag_gather(Frag<double> &f, const XV &xv)
{
    const double v = xv.at(f.c);
    f.b = f.c < 0 ? 0.0 : v;
}
template <class XV>
__device__ __forceinline__ void frag_gather(Frag<_Float16> &f, const XV &xv)
{
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const _Float16 v = xv.at(f.c[j]);
        f.b[j] = f.c[j] < 0 ? (_Float16)0 : v;
    }
}
__device__ __forceinline__ void frag_mfma(f64x4 &acc, const Frag<double> &f)
{
    acc = __builtin_amdgcn_mfma_f64_16x16x4f64(f.a, f.b, acc, 0, 0, 0);
}
__device__ __forceinline__ void frag_mfma(f32x4 &acc, const Frag<_Float16> &f)
{
    acc = __builtin_amdgcn_mfma_f32_16x16x16f16(f.a, f.b, acc, 0, 0, 0);
}

// ---- frag sources: src.load(f, i) issues the loads of the i-th MFMA step of a unit (i wave-uniform),
// src.gather(f, i, x) turns its column ids into x values once they have arrived.

// lane-linear chunks only (long pieces)
template <class T, bool NT>
struct ChunkSrc {
    static constexpr bool kPairs = false, kQuadIds = false;
    const T *val; const int *cid; size_t e0; int lane;
    template <bool PAIRED_OK = true> __device__ __forceinline__ void load(Frag<T> &f, int i) const { frag_load<NT>(f, val, cid, e0 + (size_t)i * Tr<T>::CHUNK, lane); }
    template <bool QUAD = false, class XV> __device__ __forceinline__ void gather(Frag<T> &f, int, const XV &x) const { frag_gather(f, x); }
};

// a medium block: nc lane-linear chunks, then the irregular tail as extra steps in which lane (row = l&15, kq = l>>4)
// takes the next entries of its own row.  Out-of-range lanes read element 0 of the tail arrays (never empty: the
// arena pads them) and are zeroed in gather(), so neither stage has a divergent branch.
// C8: the plan has one-byte ids (f64, 16-bit-id plans with narrow chunks): its own kernel instantiation, so that every other plan runs
// exactly the code it ran before
template <class T, bool NT, bool C16, bool PAIRS, bool C8, bool KT = false, bool REL = false>
struct BlockSrc {
    static constexpr bool kPairs = PAIRS;          // false: the windowed kernel, whose plans keep every chunk lane-linear
    static constexpr int VPL = Tr<T>::CHUNK / kWave;           // values of one chunk per lane: 1 (f64) / 4 (f16)
    ChunkSrc<T, NT> reg; int nc;
    // a pipelined block's leading chunks are stored in PAIRS, [pair][lane][2 chunks][VPL] (values and ids alike; plan.hpp med_npair):
    // the two chunks of a pair arrive with one 16-byte load per lane, which the L1 processes in as many passes as an 8-byte load
    // (4 lanes per pass) -- half the tag lookups per streamed byte (profiles/r02_pairs.md).  npair is a multiple of BATCH, so a
    // pipeline batch is either all pairs or all lane-linear chunks / tail steps.
    int npair;                                                  // chunks [0, npair) are paired
    const unsigned short *cid16; const int *base; int c0;      // C16: ids of the regular chunks as u16 offsets from base[chunk] ...
    int relb;                                                   // REL (LDS-staged window, Plan::win_rel16): every chunk's base is the window's first staged column
    // ... except the block's first n8 positions (f64: whole batches of a pipelined block's paired region): one-byte offsets, c8 = their
    // plane at the block's first element.  w16 = the u16 plane rebased so that position i's ids sit where the block's own position i
    // would be: w16 = cid16 + e16 - (e0 + n8 * CH), i.e. `w16 + at_of(i)` for i >= n8
    const unsigned char *c8; const unsigned short *w16; int n8;
    const T *ival; const int *icid; int t0, t1, kq;
    // element index of this lane's first value (and id) of regular chunk i -- wave-uniform part + lane part
    template <bool PAIRED_OK> __device__ __forceinline__ size_t at_of(int i) const
    {
        constexpr int CH = Tr<T>::CHUNK;
        if constexpr (!PAIRED_OK || !PAIRS) return reg.e0 + (size_t)i * CH + (size_t)VPL * reg.lane;       // a one-shot block: nothing is paired
        const bool paired = i < npair;
        const size_t s = reg.e0 + (paired ? (size_t)(i & ~1) * CH + (size_t)(VPL * (i & 1)) : (size_t)i * CH);
        return s + (size_t)(paired ? 2 * VPL : VPL) * reg.lane;
    }
    __device__ __forceinline__ bool pairs_ok(int i0, int n) const { return i0 + n <= npair; }
    // chunks i (even) and i + 1 of the paired region: one 16-byte load of values, one load of ids
    __device__ __forceinline__ void load2(Frag<T> &f0, Frag<T> &f1, int i) const
    {
        constexpr int CH = Tr<T>::CHUNK;
        const size_t at = reg.e0 + (size_t)i * CH + (size_t)(2 * VPL) * reg.lane;
        if constexpr (sizeof(T) == 8) {
            const f64x2 v = ldg<NT>(reinterpret_cast<const f64x2 *>(reg.val + at));
            f0.a = v[0]; f1.a = v[1];
            if constexpr (C16) {
                const unsigned r = ldg<NT>(reinterpret_cast<const unsigned *>((kQuadIds ? w16 : cid16) + at));      // raw offsets; rebased in gather() (one-shot blocks: n8 = 0)
                f0.c = (int)(r & 0xFFFFu); f1.c = (int)(r >> 16);
            } else {
                const i32x2 c = ldg<NT>(reinterpret_cast<const i32x2 *>(reg.cid + at));
                f0.c = c[0]; f1.c = c[1];
            }
        } else {
            const f16x8 v = ldg<NT>(reinterpret_cast<const f16x8 *>(reg.val + at));
            f0.a = __builtin_shufflevector(v, v, 0, 1, 2, 3); f1.a = __builtin_shufflevector(v, v, 4, 5, 6, 7);
            if constexpr (C16) {          // raw u16 offsets, two per dword; unpacked and rebased in gather()
                const i32x4 o = ldg<NT>(reinterpret_cast<const i32x4 *>(cid16 + at));
                f0.c[0] = o[0]; f0.c[1] = o[1]; f1.c[0] = o[2]; f1.c[1] = o[3];
            } else {
                f0.c = ldg<NT>(reinterpret_cast<const i32x4 *>(reg.cid + at));
                f1.c = ldg<NT>(reinterpret_cast<const i32x4 *>(reg.cid + at + 4));
            }
        }
    }
    // f64 with 16-bit ids: the four chunks i .. i + 3 of one pipeline batch inside the paired region -- two 16-byte loads of values and two
    // dword loads of ids, branch-free whether the batch is narrow (i + 4 <= n8: ONE dword per lane holds the four one-byte ids; the second
    // load repeats the first address) or wide (two pairs of u16 offsets): the base pointer is a wave-uniform select.  The raw dword stays
    // in the fragment; gather<true>() cuts the chunk's field out of it.
    static constexpr bool kQuadIds = C8 && PAIRS && C16 && sizeof(T) == 8 && Tr<T>::BATCH == 4;
    __device__ __forceinline__ void load4(Frag<T> *f, int i) const
    {
        constexpr int CH = Tr<T>::CHUNK;
        const size_t at = reg.e0 + (size_t)i * CH + (size_t)2 * reg.lane;
        const f64x2 v0 = ldg<NT>(reinterpret_cast<const f64x2 *>(reg.val + at));
        const f64x2 v1 = ldg<NT>(reinterpret_cast<const f64x2 *>(reg.val + at + 2 * CH));
        f[0].a = v0[0]; f[1].a = v0[1]; f[2].a = v1[0]; f[3].a = v1[1];
        const bool narrow = i + 4 <= n8;                        // wave-uniform
        const gbyte_p pa = narrow ? (gbyte_p)(c8 + (size_t)i * CH) : (gbyte_p)(w16 + reg.e0 + (size_t)i * CH);
        const gbyte_p pb = narrow ? pa : pa + 4 * CH;           // the wide batch's second pair: 2 chunks x CH u16 further
        const unsigned lo4 = 4u * (unsigned)reg.lane;
        const unsigned ra = ldg<NT>((const __attribute__((address_space(1))) unsigned *)(pa + lo4));
        const unsigned rb = ldg<NT>((const __attribute__((address_space(1))) unsigned *)(pb + lo4));
        f[0].c = (int)ra; f[1].c = (int)ra; f[2].c = (int)rb; f[3].c = (int)rb;
    }
    // PAIRED_OK = false: the caller knows the block has no paired chunks (the one-shot path)
    template <bool PAIRED_OK = true> __device__ __forceinline__ void load(Frag<T> &f, int i) const
    {
        if (i < nc) {
            const size_t at = at_of<PAIRED_OK>(i);
            if constexpr (!C16) frag_load_at<NT>(f, reg.val, reg.cid, at);
            else {
                if constexpr (sizeof(T) == 8) {
                    f.a = ldg<NT>(reg.val + at);
                    f.c = (int)ldg<NT>((kQuadIds ? w16 : cid16) + at);             // raw offset; rebased in gather() (single loads only see positions >= n8)
                } else {
                    f.a = ldg<NT>(reinterpret_cast<const f16x4 *>(reg.val + at));
                    const i32x2 o = ldg<NT>(reinterpret_cast<const i32x2 *>(cid16 + at));      // raw u16 offsets, two per dword
                    f.c[0] = o[0]; f.c[1] = o[1];
                }
            }
            return;
        }
        const int j = i - nc;
        if constexpr (sizeof(T) == 8) {
            const int e = t0 + 4 * j + kq;
            const int ee = e < t1 ? e : 0;
            f.a = ldg<NT>(ival + ee);
            f.c = ldg<NT>(icid + ee);
        } else {
            // the lane's 4 consecutive tail entries, element by element (a row's tail starts anywhere); entries past t1 are
            // zeroed in gather()
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int e = t0 + 16 * j + 4 * kq + q;
                const int ee = e < t1 ? e : 0;
                f.a[q] = ival[ee];
                f.c[q] = icid[ee];
            }
        }
    }
    // QUAD: the step may come from load4 (the pipelined path): inside the paired region its f.c is the batch's raw id dword
    template <bool QUAD = false, class XV> __device__ __forceinline__ void gather(Frag<T> &f, int i, const XV &x) const
    {
        if (i >= nc) {
            const int j = i - nc;
            if constexpr (sizeof(T) == 8) {
                const bool ok = t0 + 4 * j + kq < t1;
                f.a = ok ? f.a : 0.0;
                f.c = ok ? f.c : -1;
            } else {
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const bool ok = t0 + 16 * j + 4 * kq + q < t1;
                    f.a[q] = ok ? f.a[q] : (_Float16)0;
                    f.c[q] = ok ? f.c[q] : -1;
                }
            }
        } else if constexpr (C16) {
            const int b = REL ? relb : tab<KT>(base, c0 + i);   // wave-uniform: one scalar load per chunk -- or none
            if constexpr (sizeof(T) == 8) {
                unsigned o = (unsigned)f.c, pad = 0xFFFFu;
                if constexpr (QUAD && kQuadIds) {                // this chunk's field of the raw dword: wave-uniform shift / mask
                    const bool narrow = i < n8, paired = i < npair;
                    pad = narrow ? 0xFFu : 0xFFFFu;
                    o = (o >> (narrow ? 8u * (i & 3) : (paired ? 16u * (i & 1) : 0u))) & pad;
                }
                f.c = o == pad ? -1 : b + (int)o;
            }
            else {
                const unsigned lo = (unsigned)f.c[0], hi = (unsigned)f.c[1];
                const unsigned o[4] = {lo & 0xFFFFu, lo >> 16, hi & 0xFFFFu, hi >> 16};
#pragma unroll
                for (int q = 0; q < 4; ++q) f.c[q] = o[q] == 0xFFFFu ? -1 : b + (int)o[q];
            }
        }
        frag_gather(f, x);
    }
};

// the loads of the N consecutive steps from i0: pair loads when the source stores pairs and the whole batch lies in its paired
// region (one wave-uniform test per batch; i0 is then a multiple of the batch), single loads otherwise
template <int N, class SRC, class T>
__device__ __forceinline__ void load_steps(const SRC &src, Frag<T> *f, int i0)
{
    if constexpr (SRC::kPairs && N == Tr<T>::BATCH) {
        if (src.pairs_ok(i0, N)) {
            if constexpr (SRC::kQuadIds) src.load4(f, i0);
            else {
#pragma unroll
                for (int u = 0; u < N; u += 2) src.load2(f[u], f[u + 1], i0 + u);
            }
            return;
        }
    }
#pragma unroll
    for (int u = 0; u < N; ++u) src.load(f[u], i0 + u);
}

// N steps starting at step i0, everything in flight at once: all loads, then all gathers, then the MFMAs
template <class T, int N, class SRC, class ACC, class XV>
__device__ __forceinline__ void shot(ACC &acc, const SRC &src, int i0, const XV &x)
{
    Frag<T> f[N];
    if constexpr (SRC::kPairs && sizeof(T) == 2) {          // one-shot blocks: f16 pairs its chunks, f64 keeps them lane-linear (plan.hpp med_npair)
#pragma unroll
        for (int u = 0; u + 1 < N; u += 2) {
            if (src.pairs_ok(i0 + u, 2)) src.load2(f[u], f[u + 1], i0 + u);
            else { src.load(f[u], i0 + u); src.load(f[u + 1], i0 + u + 1); }
        }
        if constexpr (N % 2 == 1) src.load(f[N - 1], i0 + N - 1);
    } else if constexpr (SRC::kPairs && N >= 2) {          // f64: a one-shot block is paired as a whole (no tail steps) or not at all
        if (src.pairs_ok(i0, N & ~1)) {
#pragma unroll
            for (int u = 0; u + 1 < N; u += 2) src.load2(f[u], f[u + 1], i0 + u);
            if constexpr (N % 2 == 1) src.template load<false>(f[N - 1], i0 + N - 1);       // the odd last chunk is lane-linear
        } else {
#pragma unroll
            for (int u = 0; u < N; ++u) src.template load<false>(f[u], i0 + u);
        }
    } else {
#pragma unroll
        for (int u = 0; u < N; ++u) src.template load<false>(f[u], i0 + u);
    }
#pragma unroll
    for (int u = 0; u < N; ++u) src.gather(f[u], i0 + u, x);
#pragma unroll
    for (int u = 0; u < N; ++u) frag_mfma(acc, f[u]);
}
template <class T, int N, class SRC, class ACC, class XV>
struct ShotDispatch {
    static __device__ __forceinline__ void run(ACC &acc, const SRC &src, int i0, int n, const XV &x)
    {
        if (n == N) shot<T, N>(acc, src, i0, x);
        else ShotDispatch<T, N - 1, SRC, ACC, XV>::run(acc, src, i0, n, x);
    }
};
template <class T, class SRC, class ACC, class XV>
struct ShotDispatch<T, 0, SRC, ACC, XV> {
    static __device__ __forceinline__ void run(ACC &, const SRC &, int, int, const XV &) {}
};

// last step of the pipeline: `cur` (a full batch whose loads are in flight) and R leftover steps from i
template <class T, int U, int R, class SRC, class ACC, class XV>
struct FinishDispatch {
    static __device__ __forceinline__ void run(ACC &acc, const SRC &src, Frag<T> (&cur)[U], int ibase, int i, int rem, const XV &x)
    {
        if (rem == R) {
            Frag<T> r[R > 0 ? R : 1];
#pragma unroll
            for (int u = 0; u < U; ++u) src.template gather<true>(cur[u], ibase + u, x);
            load_steps<R>(src, r, i);
#pragma unroll
            for (int u = 0; u < U; ++u) frag_mfma(acc, cur[u]);
#pragma unroll
            for (int u = 0; u < R; ++u) src.template gather<true>(r[u], i + u, x);
#pragma unroll
            for (int u = 0; u < R; ++u) frag_mfma(acc, r[u]);
        } else if constexpr (R > 0) FinishDispatch<T, U, R - 1, SRC, ACC, XV>::run(acc, src, cur, ibase, i, rem, x);
    }
};

// All N MFMA steps of a unit.  N <= S: one shot (short blocks: a wave's whole dependent chain is
// pointers -> loads -> gathers -> MFMAs).  Longer: software-pipelined batches of U -- while batch i's x gathers are in
// flight the streaming loads of batch i+1 are already issued, so the critical path per batch is
// max(stream latency, gather latency) instead of their sum.
template <class T, int U, int S, class SRC, class ACC, class XV>
__device__ __forceinline__ void run_stream(ACC &acc, const SRC &src, int N, const XV &x)
{
    if (N <= S) { ShotDispatch<T, S, SRC, ACC, XV>::run(acc, src, 0, N, x); return; }
    const int nfull = N / U, rem = N % U;
    Frag<T> cur[U];
    load_steps<U>(src, cur, 0);
    int i = U;
    for (int it = 1; it < nfull; ++it, i += U) {
        Frag<T> nxt[U];
#pragma unroll
        for (int u = 0; u < U; ++u) src.template gather<true>(cur[u], i - U + u, x);
        load_steps<U>(src, nxt, i);
#pragma unroll
        for (int u = 0; u < U; ++u) frag_mfma(acc, cur[u]);
#pragma unroll
        for (int u = 0; u < U; ++u) cur[u] = nxt[u];
    }
    FinishDispatch<T, U, U - 1, SRC, ACC, XV>::run(acc, src, cur, i - U, i, rem, x);
}

// the one store of a row's result: y = v, or y += v in accumulate mode (wave-uniform flag; one writer per y index)
// YS (the fused multi-GPU step only): how a row's result reaches y when the own-column and the other-column product share ONE launch.
//   0: the ordinary store / read-modify-write (a.acc);
//   1: own columns -- a write-through (sc1) store, so that the value is at the coherence point once the storing wave's vmcnt wait
//      returns and the workgroup counts as done (MI355X_MICROARCH.md, inter-workgroup visibility: sc1 stores + drained + counter);
//   2: other columns, behind the in-kernel wait for ALL own-column workgroups -- sc1 load, add, sc1 store: y += v with exactly the
//      arithmetic of the two-launch form (y = own; y += other), one writer per y index in each phase.
template <class T, int YS = 0, class P>
__device__ __forceinline__ void put_y(const DevArgs &a, int yi, P v)
{
    T *y = static_cast<T *>(a.y) + yi;
    // volatile, not __hip_atomic_*: gfx950 gives a volatile access the system-scope cache bits (sc0 sc1: written through / read at the
    // coherence point), and -- unlike ANY atomic store or inline asm in the kernel -- it leaves the compiler free to fetch the row
    // tables and per-chunk bases with scalar loads (with an atomic store the own-column product of a slice runs 92 instead of 61 us)
    if constexpr (YS == 1) *(volatile T *)y = (T)v;
    else if constexpr (YS == 2) {
        const T old = *(volatile T *)y;
        *(volatile T *)y = (T)((P)old + v);
    } else *y = a.acc ? (T)((P)*y + v) : (T)v;
}

// diagonal element D[row][row] held by this lane (valid only on the 16 "diagonal lanes")
__device__ __forceinline__ bool diag_of(const f64x4 &acc, int lane, double &d)
{
    const int r = (lane & 15) >> 2;            // D reg r = row (l>>4)+4r, col l&15
    d = r == 0 ? acc[0] : r == 1 ? acc[1] : r == 2 ? acc[2] : acc[3];
    return (lane & 3) == (lane >> 4);
}
__device__ __forceinline__ bool diag_of(const f32x4 &acc, int lane, float &d)
{
    const int r = lane & 3;                    // D reg r = row 4(l>>4)+r, col l&15
    d = r == 0 ? acc[0] : r == 1 ? acc[1] : r == 2 ? acc[2] : acc[3];
    return ((lane & 15) >> 2) == (lane >> 4);
}

// ---- medium: one wave = one block of 16 sorted rows (reference: dasp_f64.h:145-279)
// YM: where the 16 results go -- 0: the block's own slots (reference permutation), or order[slot] when the plan is
// DASP_Y_NATURAL (a.order set); 2: med_dst[position] (windowed mode)
template <class T, bool NT, bool C16, int YM, bool C8 = false, int YS = 0, bool REL = false, class XV>
__device__ __forceinline__ void medium_block(const DevArgs &a, int b, int lane, const XV &x)
{
    using acc_t = typename Tr<T>::acc_t;
    constexpr int CH = Tr<T>::CHUNK;
    const T *val = static_cast<const T *>(a.med_val);
    const int c0 = tab<YS != 0>(a.med_ptr, b), c1 = tab<YS != 0>(a.med_ptr, b + 1);
    acc_t acc = {0, 0, 0, 0};
    // the block's first row is its longest (rows are sorted), so its tail length bounds the number of tail steps
    const int row = lane & 15, kq = lane >> 4;
    const int r = b * kMedRows + row;
    int t0 = 0, t1 = 0;
    if (r < a.row_block) { t0 = a.irr_ptr[r]; t1 = a.irr_ptr[r + 1]; }
    constexpr int TK = sizeof(T) == 8 ? 4 : 16;                      // tail entries of one row per MFMA step
    const int nt = (__builtin_amdgcn_readfirstlane(t1 - t0) + TK - 1) / TK;
    BlockSrc<T, NT, C16, YM != 2, C8, YS != 0, REL> src;
    if constexpr (REL) src.relb = x.cmin; else src.relb = 0;
    src.reg.val = val; src.reg.cid = a.med_cid; src.reg.e0 = (size_t)c0 * CH; src.reg.lane = lane;
    src.nc = c1 - c0; src.npair = med_npair(c1 - c0, nt, (int)sizeof(T), YM == 2 ? 0 : a.pair_mode); src.cid16 = a.med_cid16; src.base = a.med_base; src.c0 = c0;
    src.c8 = a.med_cid8; src.w16 = a.med_cid16; src.n8 = 0;
    if constexpr (C8 && C16 && sizeof(T) == 8 && YM != 2) {
        const int q0 = tab<YS != 0>(a.med_c8ptr, b), q1 = tab<YS != 0>(a.med_c8ptr, b + 1);
        src.n8 = q1 - q0; src.c8 = a.med_cid8 + (size_t)q0 * CH; src.w16 = a.med_cid16 - (size_t)q1 * CH;      // e16 - (e0 + n8 CH) = -(q0 + n8) CH
    }
    src.ival = static_cast<const T *>(a.irr_val); src.icid = a.irr_cid; src.t0 = t0; src.t1 = t1; src.kq = kq;
    run_stream<T, Tr<T>::BATCH, Tr<T>::SHOT>(acc, src, src.nc + nt, x);

    typename Tr<T>::part_t d;
    if (diag_of(acc, lane, d) && r < a.row_block) {
        const int slot = a.row_long + r;                 // row_long here = slot of the first MFMA medium row (Plan::med_slot0)
        const int yi = YM == 2 ? a.med_dst[r] : (a.order ? a.order[slot] : slot);
        put_y<T, YS>(a, yi, d);
    }
}

// ---- long: one wave = one piece (<= long_piece elements) of one long row (reference: dasp_f64.h:90-144)
template <class T, bool NT, int YS = 0>
__device__ __forceinline__ void long_piece(const DevArgs &a, int p, int lane)
{
    using acc_t = typename Tr<T>::acc_t;
    using part_t = typename Tr<T>::part_t;
    constexpr int CH = Tr<T>::CHUNK;
    constexpr int VPL = CH / kWave;          // values per lane per MFMA: 1 (f64) / 4 (f16)
    const XGlobal<T> x{static_cast<const T *>(a.x)};
    const T *val = static_cast<const T *>(a.long_val);
    const int p0 = tab<YS != 0>(a.piece_ptr, p), p1 = tab<YS != 0>(a.piece_ptr, p + 1);
    acc_t acc = {0, 0, 0, 0};
    const int full = p0 + (p1 - p0) / CH * CH;
    ChunkSrc<T, NT> src{val, a.long_cid, (size_t)p0, lane};
    run_stream<T, Tr<T>::BATCH, Tr<T>::SHOT>(acc, src, (full - p0) / CH, x);
    if (full < p1) {   // last, partial chunk: rows are padded to kLongAlign, so a lane's group is all-in or all-out;
                       // out-of-range lanes re-read the piece's first group (in bounds) and are zeroed
        const int i = full + VPL * lane;
        const bool ok = i < p1;
        Frag<T> f;
        frag_load_at<NT>(f, val, a.long_cid, (size_t)(ok ? i : p0));
        frag_gather(f, x);
        if constexpr (VPL == 1) { f.a = ok ? f.a : 0.0; f.b = ok ? f.b : 0.0; }
        else {
#pragma unroll
            for (int j = 0; j < 4; ++j) { f.a[j] = ok ? f.a[j] : (_Float16)0; f.b[j] = ok ? f.b[j] : (_Float16)0; }
        }
        frag_mfma(acc, f);
    }
    part_t d;
    const bool on_diag = diag_of(acc, lane, d);
    const part_t total = wave_sum(on_diag ? d : (part_t)0);
    if (lane == 0) {
        const int dst = tab<YS != 0>(a.piece_dst, p);
        if (dst >= 0) put_y<T, YS>(a, dst, total);
        else static_cast<part_t *>(a.partial)[~dst] = total;
    }
}

// ---- short: one wave = one tile of SHORT_ROWS rows of equal length L; lane owns V consecutive rows
template <class T, int L, bool NT, int YS = 0>
__device__ __forceinline__ void short_rows(const DevArgs &a, const ShortDev &g, int local_tile, int lane)
{
    constexpr int SR = Tr<T>::SHORT_ROWS;
    constexpr int V = SR / kWave;            // 2 (f64) / 4 (f16)
    const T *x = static_cast<const T *>(a.x);
    const T *val = static_cast<const T *>(a.short_val);
    using part_t = typename Tr<T>::part_t;
    part_t s[V];
#pragma unroll
    for (int v = 0; v < V; ++v) s[v] = 0;
    const size_t base = (size_t)g.elem_off + (size_t)local_tile * L * SR + (size_t)V * lane;
#pragma unroll
    for (int k = 0; k < L; ++k) {
        if constexpr (V == 2) {
            const f64x2 av = ldg<NT>(reinterpret_cast<const f64x2 *>(val + base + (size_t)k * SR));
            const i32x2 c = ldg<NT>(reinterpret_cast<const i32x2 *>(a.short_cid + base + (size_t)k * SR));
            const double x0 = x[c[0] < 0 ? 0 : c[0]], x1 = x[c[1] < 0 ? 0 : c[1]];   // pads: clamped gather, value dropped
            s[0] += av[0] * (c[0] < 0 ? 0.0 : x0);
            s[1] += av[1] * (c[1] < 0 ? 0.0 : x1);
        } else {
            const f16x4 av = ldg<NT>(reinterpret_cast<const f16x4 *>(val + base + (size_t)k * SR));
            const i32x4 c = ldg<NT>(reinterpret_cast<const i32x4 *>(a.short_cid + base + (size_t)k * SR));
#pragma unroll
            for (int v = 0; v < 4; ++v) {
                const _Float16 xv = x[c[v] < 0 ? 0 : c[v]];
                s[v] += (float)av[v] * (c[v] < 0 ? 0.0f : (float)xv);
            }
        }
    }
    const int t0 = local_tile * SR + V * lane;
#pragma unroll
    for (int v = 0; v < V; ++v) {
        const int t = t0 + v;
        if (t < g.count) {
            const int slot = slot_of(g.map, t);
            const int yi = a.order ? a.order[slot] : slot;
            put_y<T, YS>(a, yi, s[v]);
        }
    }
}

// the same for the medium rows stored as slabs (5 <= L <= kSlabMaxLen): L is a run-time value, four steps in flight
template <class T, bool NT, int YS = 0>
__device__ __forceinline__ void slab_rows(const DevArgs &a, const ShortDev &g, int local_tile, int lane)
{
    constexpr int SR = Tr<T>::SHORT_ROWS;
    constexpr int V = SR / kWave;
    const T *x = static_cast<const T *>(a.x);
    const T *val = static_cast<const T *>(a.short_val);
    using part_t = typename Tr<T>::part_t;
    const int L = g.len;
    part_t s[V];
#pragma unroll
    for (int v = 0; v < V; ++v) s[v] = 0;
    const size_t base = (size_t)g.elem_off + (size_t)local_tile * L * SR + (size_t)V * lane;
#pragma unroll 4
    for (int k = 0; k < L; ++k) {
        if constexpr (V == 2) {
            const f64x2 av = ldg<NT>(reinterpret_cast<const f64x2 *>(val + base + (size_t)k * SR));
            const i32x2 c = ldg<NT>(reinterpret_cast<const i32x2 *>(a.short_cid + base + (size_t)k * SR));
            const double x0 = x[c[0] < 0 ? 0 : c[0]], x1 = x[c[1] < 0 ? 0 : c[1]];
            s[0] += av[0] * (c[0] < 0 ? 0.0 : x0);
            s[1] += av[1] * (c[1] < 0 ? 0.0 : x1);
        } else {
            const f16x4 av = ldg<NT>(reinterpret_cast<const f16x4 *>(val + base + (size_t)k * SR));
            const i32x4 c = ldg<NT>(reinterpret_cast<const i32x4 *>(a.short_cid + base + (size_t)k * SR));
#pragma unroll
            for (int v = 0; v < 4; ++v) {
                const _Float16 xv = x[c[v] < 0 ? 0 : c[v]];
                s[v] += (float)av[v] * (c[v] < 0 ? 0.0f : (float)xv);
            }
        }
    }
    const int t0 = local_tile * SR + V * lane;
#pragma unroll
    for (int v = 0; v < V; ++v) {
        const int t = t0 + v;
        if (t < g.count) {
            const int slot = g.map.base[0] + t;                  // slab groups map linearly onto the medium slots
            const int yi = a.order ? a.order[slot] : slot;
            put_y<T, YS>(a, yi, s[v]);
        }
    }
}

template <class T, bool NT, int YS = 0>
__device__ __forceinline__ void short_tile(const DevArgs &a, int tile, int lane)
{
    int gi = 0;
    for (int g = 1; g < kNumShortGroups; ++g) if (tile >= a.grp_tile0[g]) gi = g;     // kernel arguments: scalar compares
    ShortDev g;
    if constexpr (YS != 0) {      // word by word through the constant address space (see tab)
        static_assert(sizeof(ShortDev) % 4 == 0, "ShortDev is a whole number of words");
        int w[sizeof(ShortDev) / 4];
#pragma unroll
        for (int i = 0; i < (int)(sizeof(ShortDev) / 4); ++i) w[i] = tab<true>(reinterpret_cast<const int *>(a.groups + gi), i);
        __builtin_memcpy(&g, w, sizeof g);
    } else g = a.groups[gi];
    const int local = tile - g.tile0;
    switch (g.len) {
        case 0: if constexpr (YS != 2) short_rows<T, 0, NT, YS>(a, g, local, lane); break;    // empty rows: y = 0 (y += 0: nothing to do)
        case 1: short_rows<T, 1, NT, YS>(a, g, local, lane); break;
        case 2: short_rows<T, 2, NT, YS>(a, g, local, lane); break;
        case 3: short_rows<T, 3, NT, YS>(a, g, local, lane); break;
        case 4: short_rows<T, 4, NT, YS>(a, g, local, lane); break;
        default: slab_rows<T, NT, YS>(a, g, local, lane); break;
    }
}

// launch bounds: the windowed kernel is held to 64 registers so that two 1024-thread window workgroups share a CU
// (A/B: 12.9 vs 15.0 us on cop20k_A); blocks are dealt to workgroups in the default round-robin order (length-sorted
// blocks in XCD-contiguous ranges put all the long ones on one XCD: DESIGN.md 4.4)
constexpr int kMinWavesPlain = 1, kMinWavesWin = 8;
// WIN: windowed mode.  A medium workgroup owns one window of row_window rows (blocks_per_win blocks, strided over its
// 4 waves); if the window's x span fits, it is copied once into LDS with coalesced 16-byte loads and every gather of
// the window reads LDS; otherwise that workgroup gathers from global memory like the non-windowed kernel.
template <class T, bool NT, bool C16, bool WIN, bool C8>
__device__ __forceinline__ void spmv_body(const DevArgs &a, char *lds_raw)
{
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int wg = blockIdx.x;
    const int wpw = WIN ? a.wpw : kWavesPerWG;
    if (wg < a.wg_long) {
        const int p = wg * wpw + wave;
        if (p < a.n_pieces) long_piece<T, NT>(a, p, lane);
    } else if (wg < a.wg_long + a.wg_med) {
        if constexpr (!WIN) {
            const int m = wg - a.wg_long;
            const XGlobal<T> x{static_cast<const T *>(a.x)};
            if (a.xcd_on) {
                // workgroups go to the XCDs round-robin, each XCD has an L2 of its own: with the blocks dealt round-robin too, every
                // XCD gathers from ALL of x (nlpkkt160: 8 x 67 MB of x through the L2s against 2.4 GB of matrix).  Rows of equal length
                // keep their row order in the sort, so a contiguous range of blocks is a contiguous part of the mesh: XCD k takes the
                // k-th eighth of the blocks (eighths of equal work) and touches an eighth of x plus the halo.
                const int k = m & 7, b = a.xcd_blk[k] + (m >> 3) * kWavesPerWG + wave;
                if (b < a.xcd_blk[k + 1]) medium_block<T, NT, C16, 0, C8>(a, b, lane, x);
            } else if (sizeof(T) == 8 && !a.med_stride) {
                // f64: the medium range is never capped (upload_plan), one block per wave -- no loop
                const int b = m * kWavesPerWG + wave;
                if (b < a.n_blocks) medium_block<T, NT, C16, 0, C8>(a, b, lane, x);
            } else {
            // grid-stride over the blocks: wg_med is capped (upload_plan) so the medium range is a persistent set of workgroups
#pragma unroll 1
            for (int b = m * kWavesPerWG + wave; b < a.n_blocks; b += a.wg_med * kWavesPerWG)
                medium_block<T, NT, C16, 0, C8>(a, b, lane, x);
            }
        } else {
            // one window per workgroup; its blocks_per_win blocks are dealt round-robin to the wpw waves.  Workgroups are dealt to the 8
            // XCDs round-robin, so workgroup m of the range takes window (m % 8) * per_xcd + m / 8: every XCD works on ONE contiguous
            // eighth of the windows -- the same eighth in every launch, whose tiles and x (an eighth of x plus the band) can stay in
            // that XCD's 4 MiB L2 from one SpMV to the next when the matrix is small enough (cop20k_A: 3.5 MB per XCD)
            const int mw = wg - a.wg_long, per_xcd = a.wg_med >> 3;
            const int w = a.win_xcd ? (mw & 7) * per_xcd + (mw >> 3) : mw;
            if (w >= a.n_windows) return;
            const int len = a.win_len[w], cmin = a.win_cmin[w];
            const T *xg = static_cast<const T *>(a.x);
            T *xw = reinterpret_cast<T *>(lds_raw);
            if (len > 0) {
                constexpr int A = 16 / (int)sizeof(T);
                const i32x4 *src = reinterpret_cast<const i32x4 *>(xg + cmin);
                i32x4 *dst = reinterpret_cast<i32x4 *>(xw);
                const int nvec = len / A, nth = wpw * kWave;
                for (int i0 = threadIdx.x; i0 < nvec; i0 += 4 * nth) {       // four 16-byte loads in flight per lane
                    const int i1 = i0 + nth, i2 = i0 + 2 * nth, i3 = i0 + 3 * nth;
                    const i32x4 v0 = src[i0];
                    const i32x4 v1 = src[i1 < nvec ? i1 : i0], v2 = src[i2 < nvec ? i2 : i0], v3 = src[i3 < nvec ? i3 : i0];
                    dst[i0] = v0;
                    if (i1 < nvec) dst[i1] = v1;
                    if (i2 < nvec) dst[i2] = v2;
                    if (i3 < nvec) dst[i3] = v3;
                }
                for (int i = nvec * A + threadIdx.x; i < len; i += nth) xw[i] = xg[cmin + i];
                __syncthreads();
                if (a.win_hybrid) {
                    const XHyb<T> x{xw, xg, cmin, (unsigned)len};
                    for (int q = wave; q < a.blocks_per_win; q += wpw) {
                        const int b = w * a.blocks_per_win + q;
                        if (b < a.n_blocks) medium_block<T, NT, C16, 2>(a, b, lane, x);
                    }
                } else if (C16 && a.win_rel16) {
                    const XLds<T> x{xw, cmin};
                    for (int q = wave; q < a.blocks_per_win; q += wpw) {
                        const int b = w * a.blocks_per_win + q;
                        if (b < a.n_blocks) medium_block<T, NT, C16, 2, false, 0, C16>(a, b, lane, x);
                    }
                } else {
                    const XLds<T> x{xw, cmin};
                    for (int q = wave; q < a.blocks_per_win; q += wpw) {
                        const int b = w * a.blocks_per_win + q;
                        if (b < a.n_blocks) medium_block<T, NT, C16, 2>(a, b, lane, x);
                    }
                }
            } else {
                const XGlobal<T> x{xg};
                for (int q = wave; q < a.blocks_per_win; q += wpw) {
                    const int b = w * a.blocks_per_win + q;
                    if (b < a.n_blocks) medium_block<T, NT, C16, 2>(a, b, lane, x);
                }
            }
        }
    } else {
        const int t = (wg - a.wg_long - a.wg_med) * wpw + wave;
        if (t < a.n_short_tiles) short_tile<T, NT>(a, t, lane);
    }
}

template <class T, bool NT, bool C16, bool WIN, bool C8 = false>
__global__ __launch_bounds__(WIN ? 1024 : 256, WIN ? kMinWavesWin : kMinWavesPlain) void dasp_spmv_kernel(DevArgs a)
{
    extern __shared__ __attribute__((aligned(16))) char lds_raw[];
    spmv_body<T, NT, C16, WIN, C8>(a, lds_raw);
}

// the windowed kernel for plans with at most one window workgroup per CU (n_windows <= CUs: cop20k_A's 212): nothing is gained by
// holding it to 64 registers for a second resident workgroup that does not exist, and at 128 it needs no scratch.
// (r3, measured and NOT kept: a wave requesting the row tables, tiles and tails of both its blocks BEFORE the x copy and the barrier,
// so that only LDS gathers and MFMAs are left behind it -- cop20k_A 10.4 -> 10.9 us.  What bounds such a workgroup is not its chain
// of latencies but its CU's memory-level parallelism: ~200 KB per CU through 64 outstanding L1 misses of ~740 cycles)
template <class T, bool C16>
__global__ __launch_bounds__(1024, 4) void dasp_spmv_win1_kernel(DevArgs a)
{
    extern __shared__ __attribute__((aligned(16))) char lds_raw[];
    spmv_body<T, false, C16, true, false>(a, lds_raw);
}

// ------------------------------------------------------------------ the fused multi-GPU step (dasp_mg_spmv, f64)
// One launch = the whole product of one rank's step.  Grid order: the workgroups of the own-column plan `a` (y = own, written
// through), then a bounded set of PERSISTENT workgroups that wait -- in the kernel -- for two flags and then stride over the
// workgroups of the other-column plan `b` (y += other: exactly the arithmetic of the two-launch form):
//   own_go   : every own-column workgroup that handles a row with other-column nonzeros ("marked", a host-built table) has stored
//              its y.  The own plan's medium blocks are dispatched through a host-built order table -- the blocks holding such rows
//              first, then the rest longest-first as ever (a plain reversal loses 20 us to the long blocks at the tail) -- so the
//              marked workgroups run FIRST and the flag is up after a fraction of the own-column product;
//   gathered : the exchange of the previous step has delivered the other ranks' x.
// So the other-column product overlaps the rest of the own-column product instead of following it, neither a kernel boundary nor a
// stream wait sits between the two, and the last workgroup to finish publishes "y ready" itself.  The waiting workgroups are at most
// max_pollers (multigpu.cpp: one slot per CU is always left free), so they can never fill the device and keep the exchange's kernel
// out; a wait longer than the time-out sets *err and skips the other-column product instead of hanging -- the host then falls back to
// the two-launch form (dasp_mg_check).
struct StepCtl {
    const unsigned long long *gathered;   // device word: step number of the last completed exchange into the gather buffer
    unsigned long long need;              // the other-column product may read the gather buffer once *gathered >= need (0: at once)
    const unsigned char *mark;            // [grid_a] 1: that own-column workgroup stores a row the other-column plan adds to
    const unsigned *mark_members;         // [64] marked workgroups per shard (wg & 63)
    int n_marked, n_mark_shards;          // marked workgroups / non-empty shards among them
    unsigned *mark_shards, *mark_top;     // arrival counters of the marked own-column workgroups (64 shards on lines of their own + top)
    unsigned long long *own_go;           // set to `step` by the last marked workgroup: the other-column product may add into y
    unsigned *all_shards, *all_top;       // arrival counters of ALL workgroups of the launch
    unsigned long long *ready;            // word the exchange waits on: set to `step` when every workgroup of this launch is done
    unsigned long long step;
    int *err;                             // sticky: 1 = a wait timed out
    int grid_a, grid_b, n_poll;           // workgroups of plan a / virtual workgroups of plan b / persistent workgroups serving them
    const int *blk_order;                 // [medium blocks of plan a] dispatch order: the blocks of marked workgroups first
    int sleep;                            // s_sleep(8) repetitions between two polls (~0.2 us each)
    long long timeout;                    // 100 MHz ticks after which a waiting workgroup gives up
};

// arrival of a workgroup at a two-level counter: true for the one that arrives last.  64 sharded counters, each on a 128-byte line
// of its own, then one top counter -- same-address atomics serialise at the memory side (one flat counter: ~50 ns per arrival,
// 180 us for the 3800 workgroups of an 8-way HV15R slice).  `members` = arrivals expected at this shard, `nshards` = non-empty
// shards.  Every counter returns to 0 with its last arrival.
constexpr int kArriveShards = 64, kShardStride = 32;        // in 4-byte words
__device__ __forceinline__ bool arrive_last(unsigned *shards, unsigned *top, int sh, unsigned members, unsigned nshards)
{
    if (__hip_atomic_fetch_add(shards + sh * kShardStride, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) + 1 != members) return false;
    __hip_atomic_store(shards + sh * kShardStride, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (__hip_atomic_fetch_add(top, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) + 1 != nshards) return false;
    __hip_atomic_store(top, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    return true;
}
// members of shard `sh` among `total` consecutively numbered arrivals
__device__ __forceinline__ unsigned shard_members(int total, int sh) { return (unsigned)((total - sh + kArriveShards - 1) / kArriveShards); }

// one (virtual) workgroup `wg` of a non-windowed plan: dasp_spmv_kernel's body, parameterised by the workgroup id.  One medium block
// per wave, no grid-stride loop (the f64 plans' medium range is never capped, upload_plan).  blk_order: the medium blocks'
// dispatch order (null: as stored).
template <class T, bool NT, bool C16, bool C8, int YS>
__device__ __forceinline__ void plain_wg(const DevArgs &a, int wg, int wave, int lane, const int *blk_order)
{
    if (wg < a.wg_long) {
        const int p = wg * kWavesPerWG + wave;
        if (p < a.n_pieces) long_piece<T, NT, YS>(a, p, lane);
    } else if (wg < a.wg_long + a.wg_med) {
        const int q = (wg - a.wg_long) * kWavesPerWG + wave;
        const XGlobal<T> x{static_cast<const T *>(a.x)};
        if (q < a.n_blocks) medium_block<T, NT, C16, 0, C8, YS>(a, blk_order ? tab<true>(blk_order, q) : q, lane, x);
    } else {
        const int t = (wg - a.wg_long - a.wg_med) * kWavesPerWG + wave;
        if (t < a.n_short_tiles) short_tile<T, NT, YS>(a, t, lane);
    }
}

template <bool NT>
__global__ __launch_bounds__(256, kMinWavesPlain) void dasp_mg_step_kernel(DevArgs a, DevArgs b, StepCtl c)
{
    __shared__ int go;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int wg = blockIdx.x;
    const int total = c.grid_a + c.n_poll;
    if (wg < c.grid_a) {
        plain_wg<double, NT, true, true, 1>(a, wg, wave, lane, c.blk_order);
        // done: every wave's write-through stores acknowledged, then ONE lane counts the workgroup.  Relaxed atomics: the y values went
        // out through sc0 sc1 stores, so an arrival needs no cache write-back or invalidate of its own (an acq_rel add costs every
        // workgroup a buffer_wbl2 + buffer_inv: measured 300 instead of 70 us per step)
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (threadIdx.x == 0) {
            if (c.n_marked > 0 && tab<true>(c.mark, wg) &&
                arrive_last(c.mark_shards, c.mark_top, wg & (kArriveShards - 1), tab<true>(c.mark_members, wg & (kArriveShards - 1)), (unsigned)c.n_mark_shards))
                __hip_atomic_store(c.own_go, c.step, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (arrive_last(c.all_shards, c.all_top, wg & (kArriveShards - 1), shard_members(total, wg & (kArriveShards - 1)),
                            (unsigned)(total < kArriveShards ? total : kArriveShards)))
                __hip_atomic_store(c.ready, c.step, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
        }
        return;
    }
    if (threadIdx.x == 0) {
        int ok = 1;
        const long long t0 = wall_clock64();
        // relaxed polls (an acquire load would invalidate caches on every iteration, under the running product), ONE acquire at the end
        while ((c.n_marked > 0 && __hip_atomic_load(c.own_go, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < c.step) ||
               (c.need && __hip_atomic_load(c.gathered, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) < c.need)) {
            for (int z = 0; z < c.sleep; ++z) __builtin_amdgcn_s_sleep(8);
            if (wall_clock64() - t0 > c.timeout) { ok = 0; __hip_atomic_store(c.err, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM); break; }
        }
        // the gather buffer was written by another kernel (this device's or, over xGMI, a peer's) while this one ran
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        go = ok;
    }
    __syncthreads();
    if (go)
        for (int v = wg - c.grid_a; v < c.grid_b; v += c.n_poll) plain_wg<double, NT, true, true, 2>(b, v, wave, lane, nullptr);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (threadIdx.x == 0 && arrive_last(c.all_shards, c.all_top, wg & (kArriveShards - 1), shard_members(total, wg & (kArriveShards - 1)),
                                        (unsigned)(total < kArriveShards ? total : kArriveShards)))
        __hip_atomic_store(c.ready, c.step, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}

// the exchange side of the fused step, on the communication stream: hold the stream until *p >= need (the product's "y ready"),
// and publish a step number behind the exchange.  Plain kernels on plain device words: no stream memory operations (Beta API).
// Neither needs a fence of its own: what the wait kernel orders is the NEXT kernel on its stream (the exchange), which acquires at its
// start like every kernel; what the flag kernel publishes was written by the PREVIOUS kernel on its stream, released at that kernel's
// end -- a release fence in a one-lane kernel is a buffer_wbl2 under the running product (rocprofv3: 6.5 us per flag kernel with it).
__global__ void dasp_mg_wait_kernel(const unsigned long long *p, unsigned long long need, long long timeout, int *err)
{
    if (threadIdx.x != 0) return;
    const long long t0 = wall_clock64();
    while (__hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) < need) {
        __builtin_amdgcn_s_sleep(8);
        if (wall_clock64() - t0 > timeout) { __hip_atomic_store(err, 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM); break; }
    }
}
__global__ void dasp_mg_flag_kernel(unsigned long long *p, unsigned long long v)
{
    if (threadIdx.x == 0) __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}

// stage 2 for long rows cut into several pieces (reference: longPart_sum, dasp_f64.h:53-75)
template <class T>
__global__ __launch_bounds__(256) void dasp_long_reduce_kernel(DevArgs a)
{
    using part_t = typename Tr<T>::part_t;
    const int lane = threadIdx.x & 63;
    const int i = blockIdx.x * kWavesPerWG + (threadIdx.x >> 6);
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

Plan::~Plan()
{
    if (dev) {
        if (dev->arena) (void)hipFree(dev->arena);
        delete dev;
    }
}

static int require_device()
{
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess || n <= 0) {
        set_error("no HIP device visible (the DASP GPU path has no CPU fallback)");
        return DASP_ERR_NO_DEVICE;
    }
    return DASP_OK;
}

int upload_plan(Plan &p);
static int upload_plan_impl(Plan &p)
{
    if (int rc = require_device()) return rc;
    if (p.host_dropped && p.dev) return DASP_OK;   // already on the device (packed there, or host copies released)
    if (p.host_dropped) { set_error("host arrays were dropped"); return DASP_ERR_STATE; }
    if (p.dev) { if (p.dev->arena) (void)hipFree(p.dev->arena); delete p.dev; p.dev = nullptr; }
    auto *d = new DevicePlan();
    p.dev = d;
    HIP_TRY(hipGetDevice(&d->device));
    if (!p.panels.empty()) {   // column panels: every panel is a plan of its own; this one only owns their partial results
        for (auto &h : p.panels) if (int rc = upload_plan(h->impl)) return rc;
        d->ypart_stride = ((size_t)std::max(p.m, 1) + 127) & ~size_t(127);
        d->arena_bytes = d->ypart_stride * p.panels.size() * (size_t)p.geo.vbytes;
        HIP_TRY(hipMalloc(&d->arena, d->arena_bytes));
        return DASP_OK;
    }

    std::vector<ShortDev> groups(kNumShortGroups);
    for (int g = 0; g < kNumShortGroups; ++g) {
        groups[g].len = p.grp[g].len; groups[g].count = p.grp[g].count; groups[g].tiles = p.grp[g].tiles;
        groups[g].tile0 = p.grp[g].tile0; groups[g].elem_off = p.grp[g].elem_off; groups[g].map = p.grp[g].map;
    }
    const bool natural = p.opt.y_order == DASP_Y_NATURAL;
    const size_t part_bytes = (size_t)std::max<size_t>(1, p.multi_ptr.empty() ? 0 : (size_t)p.multi_ptr.back()) * 8;

    struct Item { const void *src; size_t bytes; size_t off; };
    std::vector<Item> items;
    size_t total = 0;
    auto add = [&](const void *src, size_t bytes) {
        size_t off = total;
        items.push_back({src, bytes, off});
        total += (bytes + 255) & ~size_t(255);
        if (bytes == 0) total += 256;
        return off;
    };
    // nnz-sized arrays: sized by their element counts; a plan packed on the device has no host copy (src = nullptr)
    auto src_of = [](const auto &v) -> const void * { return v.empty() ? nullptr : v.data(); };
    const size_t vbytes = (size_t)p.geo.vbytes;
    const size_t o_lv = add(src_of(p.long_val), p.cnt_long * vbytes);
    const size_t o_lc = add(src_of(p.long_cid), p.cnt_long * 4);
    const size_t o_pp = add(p.piece_ptr.data(), p.piece_ptr.size() * 4);
    const size_t o_pd = add(p.piece_dst.data(), p.piece_dst.size() * 4);
    const size_t o_mp = add(p.multi_ptr.data(), p.multi_ptr.size() * 4);
    const size_t o_md = add(p.multi_dst.data(), p.multi_dst.size() * 4);
    const size_t o_part = add(nullptr, part_bytes);
    const size_t o_mptr = add(p.med_ptr.data(), p.med_ptr.size() * 4);
    const size_t o_mv = add(src_of(p.med_val), p.cnt_reg * vbytes);
    const size_t o_mc = add(src_of(p.med_cid), p.cid16 ? 0 : p.cnt_reg * 4);
    const size_t o_mc16 = add(src_of(p.med_cid16), p.cid16 ? (p.cnt_reg - p.cnt_reg8) * 2 : 0);
    const size_t o_mc8 = add(src_of(p.med_cid8), p.cnt_reg8);
    const size_t o_c8p = add(p.med_c8ptr.data(), p.med_c8ptr.size() * 4);
    const size_t o_mb = add(src_of(p.med_base), p.cid16 ? (size_t)p.med_ptr.back() * 4 : 0);
    const size_t o_ip = add(p.irr_ptr.data(), p.irr_ptr.size() * 4);
    const size_t o_iv = add(src_of(p.irr_val), p.cnt_irr * vbytes);
    const size_t o_ic = add(src_of(p.irr_cid), p.cnt_irr * 4);
    const size_t o_mdst = add(p.med_dst.data(), p.med_dst.size() * 4);
    const size_t o_wc = add(p.win_cmin.data(), p.win_cmin.size() * 4);
    const size_t o_wl = add(p.win_len.data(), p.win_len.size() * 4);
    const size_t o_sv = add(src_of(p.short_val), p.cnt_short * vbytes);
    const size_t o_sc = add(src_of(p.short_cid), p.cnt_short * 4);
    const size_t o_g = add(groups.data(), groups.size() * sizeof(ShortDev));
    std::vector<int> ord_mapped;   // a column panel writes row r to dst_map[r] (its parent's slot), not to r
    if (natural && !p.dst_map.empty()) { ord_mapped.resize(p.order.size()); for (size_t i = 0; i < p.order.size(); ++i) ord_mapped[i] = p.dst_map[(size_t)p.order[i]]; }
    const size_t o_ord = add(natural ? (ord_mapped.empty() ? p.order.data() : ord_mapped.data()) : nullptr, natural ? p.order.size() * 4 : 0);

    HIP_TRY(hipMalloc(&d->arena, total));
    d->arena_bytes = total;
    if (std::getenv("DASP_VERBOSE")) std::fprintf(stderr, "[dasp upload] arena %p + %zu bytes\n", d->arena, total);
    char *base = static_cast<char *>(d->arena);
    for (const Item &it : items)
        if (it.src && it.bytes) HIP_TRY(hipMemcpy(base + it.off, it.src, it.bytes, hipMemcpyHostToDevice));

    d->map.long_val = o_lv; d->map.long_cid = o_lc; d->map.med_val = o_mv; d->map.med_cid = o_mc; d->map.med_cid16 = o_mc16; d->map.med_cid8 = o_mc8;
    d->map.med_base = o_mb; d->map.irr_val = o_iv; d->map.irr_cid = o_ic; d->map.short_val = o_sv; d->map.short_cid = o_sc;
    DevArgs &a = d->args;
    a.long_val = base + o_lv; a.long_cid = (const int *)(base + o_lc);
    a.piece_ptr = (const int *)(base + o_pp); a.piece_dst = (const int *)(base + o_pd);
    a.multi_ptr = (const int *)(base + o_mp); a.multi_dst = (const int *)(base + o_md);
    a.partial = base + o_part;
    a.n_pieces = (int)p.piece_dst.size(); a.n_multi = (int)p.multi_dst.size();
    a.med_ptr = (const int *)(base + o_mptr); a.med_val = base + o_mv; a.med_cid = (const int *)(base + o_mc);
    a.irr_ptr = (const int *)(base + o_ip); a.irr_val = base + o_iv; a.irr_cid = (const int *)(base + o_ic);
    a.n_blocks = p.stats.n_med_blocks; a.row_block = p.n_mfma_rows; a.row_long = p.med_slot0;
    for (int g = 0; g < kNumShortGroups; ++g) a.grp_tile0[g] = p.grp[g].tile0;
    a.short_val = base + o_sv; a.short_cid = (const int *)(base + o_sc); a.groups = (const ShortDev *)(base + o_g);
    a.n_short_tiles = p.stats.n_short_tiles;
    a.order = natural ? (const int *)(base + o_ord) : nullptr;
    a.wpw = p.windowed ? std::min(16, p.row_window / kMedRows) : kWavesPerWG;
    a.wg_long = (a.n_pieces + a.wpw - 1) / a.wpw;
    a.med_cid16 = (const unsigned short *)(base + o_mc16); a.med_base = (const int *)(base + o_mb);
    a.med_cid8 = (const unsigned char *)(base + o_mc8); a.med_c8ptr = (const int *)(base + o_c8p);
    a.med_dst = (const int *)(base + o_mdst); a.win_cmin = (const int *)(base + o_wc); a.win_len = (const int *)(base + o_wl);
    a.n_windows = (int)p.win_len.size(); a.blocks_per_win = p.windowed ? p.row_window / kMedRows : 0;
    a.win_hybrid = p.win_hybrid ? 1 : 0; a.win_rel16 = p.win_rel16 ? 1 : 0; a.pair_mode = p.pair_mode;
    a.win_xcd = 1;
    if (const char *e = std::getenv("DASP_WIN_XCD")) a.win_xcd = std::atoi(e);      // A/B knob
    d->win1 = false;
    if (p.windowed) {
        int cus = 256;
        hipDeviceProp_t prop;
        if (hipGetDeviceProperties(&prop, d->device) == hipSuccess && prop.multiProcessorCount > 0) cus = prop.multiProcessorCount;
        d->win1 = a.n_windows <= cus;
        if (const char *e = std::getenv("DASP_WIN1")) d->win1 = d->win1 && std::atoi(e) != 0;
    }
    a.wg_med = p.windowed ? (a.n_windows + 7) / 8 * 8 : (a.n_blocks + kWavesPerWG - 1) / kWavesPerWG;      // windows: a whole number per XCD (kernel)
    // f16 blocks of uniform length: a persistent set of 7 workgroups per CU striding over the blocks amortises the per-wave
    // set-up that weighs twice as much at 2 bytes per value (nlpkkt160 f16 0.675 -> 0.739 of the roofline, Queen_4147 f16
    // 0.873 -> 0.927).  Static striding needs equal blocks: with HV15R's 2 % of 3x longer rows it loses 10 %, and in f64 it
    // loses 3-9 % everywhere, so: f16 only, no windows, longest block <= 1.25 x the mean.
    if (!p.windowed && p.precision == 16 && a.n_blocks > 256 * 7 * kWavesPerWG) {      // (the threshold: a full set on a 256-CU device)
        int longest = 0;
        for (int b = 0; b < a.n_blocks; ++b) longest = std::max(longest, p.med_ptr[(size_t)b + 1] - p.med_ptr[(size_t)b]);
        const double mean = (double)p.med_ptr[(size_t)a.n_blocks] / (double)a.n_blocks;
        if (p.cnt_irr * 8 <= p.cnt_reg && mean > 0 && (double)longest <= 1.25 * mean) {
            // as many persistent workgroups per CU as the kernel's registers let reside at once (7 at <= 72 VGPRs), on every CU of the device
            int per_cu = 7, cus = 256;
            hipDeviceProp_t prop;
            if (hipGetDeviceProperties(&prop, d->device) == hipSuccess && prop.multiProcessorCount > 0) cus = prop.multiProcessorCount;
            const void *fn = p.cid16 ? reinterpret_cast<const void *>(&dasp_spmv_kernel<_Float16, true, true, false>)
                                     : reinterpret_cast<const void *>(&dasp_spmv_kernel<_Float16, true, false, false>);
            int fit = 0;
            if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&fit, fn, kWave * kWavesPerWG, 0) == hipSuccess && fit > 0) per_cu = std::min(per_cu, fit);
            (void)hipGetLastError();
            a.wg_med = std::min(a.wg_med, cus * per_cu);
        }
    }
    // XCD-contiguous block ranges (opt-in: DASP_XCD_BLOCKS=1) for plans of EVEN blocks (FEM / stencil rows: longest block <= 4 x the
    // mean).  Work of a block = its chunks + 2 (row tables, tail); every XCD gets an eighth of it.  Measured r3 (profiles/r03_xcd_blocks.md):
    // it removes exactly the traffic it was built for -- nlpkkt160's x pulled through eight L2s, FETCH 3.03 -> 2.54 GB per SpMV, 1.03 ->
    // 0.86 x the CSR bytes -- but that traffic was Infinity-Cache hits, not HBM reads, and the time does not move (nlpkkt160 0.4051 ->
    // 0.4105 ms, HV15R 0.4274 -> 0.4227, Queen 0.5089 -> 0.5083, f16 0-1.5 % slower), so it stays off by default.
    a.xcd_on = 0;
    for (int &v : a.xcd_blk) v = 0;
    a.med_stride = a.wg_med * kWavesPerWG < a.n_blocks ? 1 : 0;      // the medium range is a persistent, striding set of workgroups (f16 only today)
    if (const char *e = std::getenv("DASP_MED_LOOP")) a.med_stride = a.med_stride || std::atoi(e) != 0;      // A/B knob
    if (!p.windowed && a.wg_med == (a.n_blocks + kWavesPerWG - 1) / kWavesPerWG && a.n_blocks >= 8 * 256 && (int)p.med_ptr.size() == a.n_blocks + 1) {
        int longest = 0;
        for (int b = 0; b < a.n_blocks; ++b) longest = std::max(longest, p.med_ptr[(size_t)b + 1] - p.med_ptr[(size_t)b]);
        const double mean = (double)p.med_ptr[(size_t)a.n_blocks] / (double)a.n_blocks;
        const char *e = std::getenv("DASP_XCD_BLOCKS");
        const bool on = e && std::atoi(e) != 0 && (double)longest <= 4.0 * std::max(mean, 1.0);
        if (on) {
            auto work_before = [&](int b) { return (long long)p.med_ptr[(size_t)b] + 2ll * b; };
            const long long total = work_before(a.n_blocks);
            int most = 0;
            for (int k = 0; k <= 8; ++k) {
                int lo = 0, hi = a.n_blocks;                               // first block whose preceding work reaches k / 8 of the total
                while (lo < hi) { const int mid = (lo + hi) / 2; if (work_before(mid) * 8 < total * k) lo = mid + 1; else hi = mid; }
                a.xcd_blk[k] = k == 8 ? a.n_blocks : lo;
            }
            for (int k = 0; k < 8; ++k) most = std::max(most, a.xcd_blk[k + 1] - a.xcd_blk[k]);
            a.xcd_on = 1;
            a.wg_med = 8 * ((most + kWavesPerWG - 1) / kWavesPerWG);
        }
    }
    a.wg_short = (a.n_short_tiles + a.wpw - 1) / a.wpw;
    // streamed-once matrix data bypasses the caches (the reference's ld.global.cs, dasp_f64.h:34-51)
    // only when it cannot stay resident in the 256 MiB Infinity Cache between two SpMVs anyway.
    d->nt = p.opt.stream_policy == 2 || (p.opt.stream_policy != 1 && p.stats.data_X > kStreamBytes);
    if (p.windowed && p.lds_bytes > 65536) {
        // more than the default 64 KiB of dynamic LDS must be requested per kernel; done here (for both cache-policy
        // variants), not in the launch path, so that dasp_plan_spmv stays free of anything a stream capture would reject
        // the attribute belongs to the kernel, not to the plan: always ask for the device maximum, or a later plan with narrower
        // windows would lower the limit under an earlier one with wider windows
        const int bytes = 160 * 1024;
        const bool c16 = p.cid16;
        hipError_t e1, e2;
#define DASP_ATTR(TT, NTV, CV) hipFuncSetAttribute(reinterpret_cast<const void *>(&dasp_spmv_kernel<TT, NTV, CV, true>), hipFuncAttributeMaxDynamicSharedMemorySize, bytes)
        if (p.precision == 64) { e1 = c16 ? DASP_ATTR(double, true, true) : DASP_ATTR(double, true, false); e2 = c16 ? DASP_ATTR(double, false, true) : DASP_ATTR(double, false, false); }
        else { e1 = c16 ? DASP_ATTR(_Float16, true, true) : DASP_ATTR(_Float16, true, false); e2 = c16 ? DASP_ATTR(_Float16, false, true) : DASP_ATTR(_Float16, false, false); }
#undef DASP_ATTR
        HIP_TRY(e1);
        HIP_TRY(e2);
        if (p.precision == 64) HIP_TRY(c16 ? hipFuncSetAttribute(reinterpret_cast<const void *>(&dasp_spmv_win1_kernel<double, true>), hipFuncAttributeMaxDynamicSharedMemorySize, bytes)
                                           : hipFuncSetAttribute(reinterpret_cast<const void *>(&dasp_spmv_win1_kernel<double, false>), hipFuncAttributeMaxDynamicSharedMemorySize, bytes));
        else HIP_TRY(c16 ? hipFuncSetAttribute(reinterpret_cast<const void *>(&dasp_spmv_win1_kernel<_Float16, true>), hipFuncAttributeMaxDynamicSharedMemorySize, bytes)
                         : hipFuncSetAttribute(reinterpret_cast<const void *>(&dasp_spmv_win1_kernel<_Float16, false>), hipFuncAttributeMaxDynamicSharedMemorySize, bytes));
    }
    return DASP_OK;
}

int launch_spmv(Plan &p, const void *dX, void *dY, void *stream, bool accumulate);

// ---- placement trials (r3, profiles/r03_placement.md).  The same arena bytes run the HBM-bound kernels at one of two speeds ~8 % apart
// depending on WHERE the allocation landed (two uploads of one plan in one process, interleaved timing: 0.420 vs 0.453 ms; the skew
// between the arena's arrays, the TLB and the history of the device do not matter; which allocation of a process is the fast one differs
// from box to box).  A user-mode library cannot see the cause, but it can look: time a few launches, copy the arena into a fresh allocation
// (the old one stays allocated meanwhile, so the new one lands elsewhere), time again, keep the faster, at most `trials` allocations alive at once
// (DASP_PLACEMENT_TRIALS, default 6; 1 = off), stop as soon as one allocation is >= 4 % faster than another.  Only plans that stream
// >= 256 MiB per SpMV and are not gather-bound by construction (column panels, LDS windows).  Costs ~5 ms per trial for HV15R.
// Measured on the bench headline, six fresh processes each on one box: without 0.4343 0.4272 0.4595 0.4595 0.4603 0.4594 ms, with
// 0.4508 0.4283 0.4285 0.4275 0.4332 0.4336 ms -- the caller's x / y take part in the effect (profiles/r03_placement.md), so the trials
// with scratch operands shift the odds, they do not decide.
static void rebase_args(DevArgs &a, const char *from, const char *to, size_t bytes)
{
    auto mv = [&](auto &ptr) {
        const char *q = reinterpret_cast<const char *>(ptr);
        if (q >= from && q < from + bytes) ptr = reinterpret_cast<std::remove_reference_t<decltype(ptr)>>(const_cast<char *>(to + (q - from)));
    };
    mv(a.long_val); mv(a.long_cid); mv(a.piece_ptr); mv(a.piece_dst); mv(a.partial); mv(a.multi_ptr); mv(a.multi_dst);
    mv(a.med_ptr); mv(a.med_val); mv(a.med_cid); mv(a.med_cid16); mv(a.med_base); mv(a.med_cid8); mv(a.med_c8ptr);
    mv(a.irr_ptr); mv(a.irr_val); mv(a.irr_cid); mv(a.med_dst); mv(a.win_cmin); mv(a.win_len);
    mv(a.short_val); mv(a.short_cid); mv(a.groups); mv(a.order);
}

int tune_placement(Plan &p, int trials, const void *dX, void *dY, double *ms_first, double *ms_kept)
{
    DevicePlan *d = p.dev;
    if (ms_first) *ms_first = 0.0;
    if (ms_kept) *ms_kept = 0.0;
    if (trials <= 0) {
        trials = 6;
        if (const char *e = std::getenv("DASP_PLACEMENT_TRIALS")) trials = std::max(1, std::min(8, std::atoi(e)));
    }
    if (!d || !d->arena || trials <= 1 || d->arena_bytes < (size_t(256) << 20) || !p.panels.empty() || p.panel || p.windowed) return DASP_OK;
    const bool verbose = std::getenv("DASP_VERBOSE") != nullptr;
    const size_t vb = (size_t)p.geo.vbytes, bytes = d->arena_bytes;
    const size_t xlen = p.opt.n_parts > 0 ? (size_t)p.opt.n_parts * (size_t)p.opt.part_stride : (size_t)p.n;
    void *x = nullptr, *y = nullptr;                 // scratch operands unless the caller lends its own (whose placement takes part in the effect)
    const void *ux = dX; void *uy = dY;
    hipEvent_t e0 = nullptr, e1 = nullptr;
    std::vector<void *> losers;
    auto cleanup = [&] {
        for (void *q : losers) (void)hipFree(q);
        if (x) (void)hipFree(x);
        if (y) (void)hipFree(y);
        if (e0) (void)hipEventDestroy(e0);
        if (e1) (void)hipEventDestroy(e1);
        (void)hipGetLastError();
    };
    // scratch operands: zeros (the values do not matter to the stream); a failure anywhere below leaves the plan as it is
    if (!ux) { if (hipMalloc(&x, std::max<size_t>(xlen * vb, 256)) != hipSuccess || hipMemset(x, 0, std::max<size_t>(xlen * vb, 256)) != hipSuccess) { cleanup(); return DASP_OK; } ux = x; }
    if (!uy) { if (hipMalloc(&y, ((size_t)p.m + 64) * vb) != hipSuccess) { cleanup(); return DASP_OK; } uy = y; }
    if (hipEventCreate(&e0) != hipSuccess || hipEventCreate(&e1) != hipSuccess) { cleanup(); return DASP_OK; }
    auto time_it = [&](double *ms) -> bool {
        for (int i = 0; i < 2; ++i) if (launch_spmv(p, ux, uy, nullptr, false) != DASP_OK) return false;
        if (hipEventRecord(e0, nullptr) != hipSuccess) return false;
        const int reps = 6;
        for (int i = 0; i < reps; ++i) if (launch_spmv(p, ux, uy, nullptr, false) != DASP_OK) return false;
        float t = 0.f;
        if (hipEventRecord(e1, nullptr) != hipSuccess || hipEventSynchronize(e1) != hipSuccess || hipEventElapsedTime(&t, e0, e1) != hipSuccess) return false;
        *ms = (double)t / reps;
        return true;
    };
    double best = 0.0;
    if (!time_it(&best)) { cleanup(); return DASP_OK; }
    double lo = best, hi = best;
    if (ms_first) *ms_first = best;
    if (verbose) std::fprintf(stderr, "[dasp placement] allocation 0 at %p: %.4f ms\n", d->arena, best);
    for (int t = 1; t < trials && hi < 1.04 * lo; ++t) {
        void *na = nullptr;
        if (hipMalloc(&na, bytes) != hipSuccess) { (void)hipGetLastError(); break; }
        if (hipMemcpy(na, d->arena, bytes, hipMemcpyDeviceToDevice) != hipSuccess) { (void)hipFree(na); (void)hipGetLastError(); break; }
        char *old = static_cast<char *>(d->arena);
        rebase_args(d->args, old, static_cast<char *>(na), bytes);
        d->arena = na;
        double ms = 0.0;
        const bool ok = time_it(&ms);
        if (verbose) std::fprintf(stderr, "[dasp placement] allocation %d at %p: %.4f ms\n", t, na, ok ? ms : -1.0);
        if (ok && ms < best) { best = ms; losers.push_back(old); }
        else {                                        // back to the one that was faster
            rebase_args(d->args, static_cast<char *>(na), old, bytes);
            d->arena = old;
            losers.push_back(na);
        }
        if (ok) { lo = std::min(lo, ms); hi = std::max(hi, ms); }
    }
    if (hipDeviceSynchronize() != hipSuccess) (void)hipGetLastError();
    if (ms_kept) *ms_kept = best;
    // the allocations that lost go back now -- and the driver wipes released VRAM in the background, which costs the kernels 1-3 % for the
    // next 50-500 ms, by how much was released (seen as a 2.8 % slower timed region right behind six trials on a box where none of them was
    // faster).  Let that pass here, at set-up, not under the caller's first products: launches until they run as fast as the kept
    // allocation did.
    const size_t freed_bytes = losers.size() * bytes;
    const bool freed = !losers.empty();
    for (void *q : losers) (void)hipFree(q);
    losers.clear();
    if (freed) {
        // ~35 GB/s of wiping was seen (5.3 GB: 0.15 s of slower launches); wait for 20 GB/s worth, then check with launches
        std::this_thread::sleep_for(std::chrono::duration<double>(std::min(1.0, (double)freed_bytes / 20e9)));
        const auto t0 = std::chrono::steady_clock::now();
        double ms = 0.0, prev = 0.0;
        int rounds = 0, steady = 0;
        // done when the kernel is back at the kept allocation's speed, or has stopped changing (three rounds within 0.3 % of each other)
        while (time_it(&ms) && std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() < 0.5) {
            ++rounds;
            if (ms <= 1.004 * best) break;
            steady = prev > 0.0 && std::fabs(ms - prev) <= 0.003 * ms ? steady + 1 : 0;
            if (steady >= 3) break;
            prev = ms;
        }
        if (verbose) std::fprintf(stderr, "[dasp placement] settled after %d more rounds (%.4f ms)\n", rounds, ms);
    }
    cleanup();
    return DASP_OK;
}

// host-built plans: the trials are part of the upload (a few ms next to the packing).  Plans packed on the device
// (dasp_plan_create_device: creation time is the metric there) leave them to the caller: dasp_plan_tune_placement.
int upload_plan(Plan &p)
{
    const bool fresh = !(p.host_dropped && p.dev);
    if (int rc = upload_plan_impl(p)) return rc;
    return fresh ? tune_placement(p, 0, nullptr, nullptr, nullptr, nullptr) : DASP_OK;
}
// the arena with every O(rows) array, the nnz-sized regions left for the device packers (devpack.hip)
int upload_plan_unpacked(Plan &p) { return upload_plan_impl(p); }

template <class T>
static int launch_typed(Plan &p, const DevArgs &a, hipStream_t s)
{
    const int grid = a.wg_long + a.wg_med + a.wg_short;
    const bool nt = p.dev->nt;
    if (grid > 0) {
        const size_t lds = p.windowed ? (size_t)p.lds_bytes : 0;
        const bool c16 = p.cid16;
#define DASP_FOR_EACH(M) \
        if (nt && c16 && p.windowed) { M(true, true, true); } else if (nt && c16) { M(true, true, false); } \
        else if (nt && p.windowed) { M(true, false, true); } else if (nt) { M(true, false, false); } \
        else if (c16 && p.windowed) { M(false, true, true); } else if (c16) { M(false, true, false); } \
        else if (p.windowed) { M(false, false, true); } else { M(false, false, false); }
#define DASP_LAUNCH(NTV, CV, WINV) hipLaunchKernelGGL((dasp_spmv_kernel<T, NTV, CV, WINV>), dim3(grid), dim3(kWave * a.wpw), lds, s, a)
        if (p.windowed && p.dev->win1 && !nt) {                             // at most one window workgroup per CU: the 128-register build
            if (c16) hipLaunchKernelGGL((dasp_spmv_win1_kernel<T, true>), dim3(grid), dim3(kWave * a.wpw), lds, s, a);
            else hipLaunchKernelGGL((dasp_spmv_win1_kernel<T, false>), dim3(grid), dim3(kWave * a.wpw), lds, s, a);
        } else if (sizeof(T) == 8 && c16 && !p.windowed && p.cnt_reg8 > 0) {      // plans with one-byte ids: their own instantiation
            if (nt) hipLaunchKernelGGL((dasp_spmv_kernel<double, true, true, false, true>), dim3(grid), dim3(kWave * a.wpw), lds, s, a);
            else hipLaunchKernelGGL((dasp_spmv_kernel<double, false, true, false, true>), dim3(grid), dim3(kWave * a.wpw), lds, s, a);
        } else
        DASP_FOR_EACH(DASP_LAUNCH)
#undef DASP_LAUNCH
#undef DASP_FOR_EACH
    }
    if (a.n_multi > 0)
        hipLaunchKernelGGL((dasp_long_reduce_kernel<T>), dim3((a.n_multi + kWavesPerWG - 1) / kWavesPerWG), dim3(256), 0, s, a);
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

int launch_spmv(Plan &p, const void *dX, void *dY, void *stream, bool accumulate)
{
    if (!p.dev || !p.dev->arena) { set_error("plan not uploaded"); return DASP_ERR_STATE; }
    if (!dX || !dY) { set_error("null device pointer"); return DASP_ERR_ARG; }
    if (!p.panels.empty()) {
        const size_t vb = (size_t)p.geo.vbytes, stride = p.dev->ypart_stride;
        char *part = static_cast<char *>(p.dev->arena);
        for (size_t k = 0; k < p.panels.size(); ++k)
            if (int rc = launch_spmv(p.panels[k]->impl, dX, part + k * stride * vb, stream, false)) return rc;
        hipStream_t s = static_cast<hipStream_t>(stream);
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
    hipStream_t s = static_cast<hipStream_t>(stream);
    return p.precision == 64 ? launch_typed<double>(p, a, s) : launch_typed<_Float16>(p, a, s);
}

// ---- fused multi-GPU step (multigpu.cpp).  Which plans qualify: f64, uploaded, no x windows, no column panels, 16-bit ids (the one
// instantiation of the step kernel), no long row cut into several pieces (their stage 2 would run behind the launch that publishes
// "y ready").  `other` may be null (no nonzero outside the rank's own columns).
bool mg_step_supported(const Plan &own, const Plan *other)
{
    auto ok = [](const Plan &p) {
        // 16-bit ids -- or no MFMA block at all (every medium row stored as a slab: nothing reads the id planes)
        return p.precision == 64 && p.dev && p.dev->arena && p.panels.empty() && !p.windowed && (p.cid16 || p.stats.n_med_blocks == 0) && p.dev->args.n_multi == 0;
    };
    return ok(own) && (!other || ok(*other));
}

void mg_step_marks(const Plan &p, const unsigned char *has_other, std::vector<unsigned char> &mark, std::vector<int> &blk_order)
{
    // the launch grid of upload_plan / dasp_mg_step_kernel: [ long pieces | medium blocks through blk_order | short tiles ], 4 units per workgroup
    const int n_pieces = (int)p.piece_dst.size(), n_blocks = p.stats.n_med_blocks, n_tiles = p.stats.n_short_tiles;
    const int wg_long = (n_pieces + kWavesPerWG - 1) / kWavesPerWG, wg_med = (n_blocks + kWavesPerWG - 1) / kWavesPerWG,
              wg_short = (n_tiles + kWavesPerWG - 1) / kWavesPerWG;
    mark.assign((size_t)wg_long + wg_med + wg_short, 0);
    for (int q = 0; q < n_pieces; ++q) {
        const int dst = p.piece_dst[(size_t)q];
        if (dst >= 0 && has_other[dst]) mark[(size_t)q / kWavesPerWG] = 1;
    }
    // medium blocks holding a row the other-column plan adds to go first, everything else keeps the stored (longest-first) order
    std::vector<unsigned char> hot((size_t)n_blocks, 0);
    for (int b = 0; b < n_blocks; ++b)
        for (int i = 0; i < kMedRows && b * kMedRows + i < p.n_mfma_rows; ++i)
            if (has_other[p.order[(size_t)p.med_slot0 + (size_t)b * kMedRows + i]]) { hot[(size_t)b] = 1; break; }
    blk_order.clear(); blk_order.reserve((size_t)n_blocks);
    for (int b = 0; b < n_blocks; ++b) if (hot[(size_t)b]) blk_order.push_back(b);
    const int n_hot = (int)blk_order.size();
    for (int b = 0; b < n_blocks; ++b) if (!hot[(size_t)b]) blk_order.push_back(b);
    for (int q = 0; q < n_hot; ++q) mark[(size_t)wg_long + q / kWavesPerWG] = 1;
    const int SR = p.geo.short_rows;
    for (int g = 0; g < kNumShortGroups; ++g) {
        const ShortGroup &G = p.grp[g];
        for (int lt = 0; lt < G.tiles; ++lt) {
            const int t = G.tile0 + lt;
            for (int tt = lt * SR; tt < std::min(G.count, (lt + 1) * SR); ++tt) {
                const int slot = g < 5 ? G.map.slot(tt) : G.map.base[0] + tt;      // short_rows / slab_rows
                if (has_other[p.order[(size_t)slot]]) { mark[(size_t)wg_long + wg_med + t / kWavesPerWG] = 1; break; }
            }
        }
    }
}

int launch_mg_step(Plan &own, Plan *other, const void *x_own, const void *x_gathered, void *y, const MgStepCtl &h, void *stream)
{
    if (!mg_step_supported(own, other)) { set_error("plans do not qualify for the fused multi-GPU step"); return DASP_ERR_STATE; }
    DevArgs a = own.dev->args, b = other ? other->dev->args : own.dev->args;
    a.x = x_own; a.y = y; a.acc = 0;
    b.x = x_gathered; b.y = y; b.acc = 0;
    // the step kernel's own geometry: one medium block per wave in table order (mg_step_marks assumes it), whatever upload_plan chose
    a.wg_med = (a.n_blocks + kWavesPerWG - 1) / kWavesPerWG; a.xcd_on = 0;
    b.wg_med = (b.n_blocks + kWavesPerWG - 1) / kWavesPerWG; b.xcd_on = 0;
    StepCtl c{};
    char *w = static_cast<char *>(h.words);
    c.mark_shards = reinterpret_cast<unsigned *>(w); c.all_shards = reinterpret_cast<unsigned *>(w + 8192);
    c.mark_top = reinterpret_cast<unsigned *>(w + 16384); c.all_top = reinterpret_cast<unsigned *>(w + 16384 + 256);
    c.gathered = reinterpret_cast<const unsigned long long *>(w + kMgWordGathered); c.need = other ? h.need : 0;
    c.own_go = reinterpret_cast<unsigned long long *>(w + kMgWordOwnGo);
    c.ready = reinterpret_cast<unsigned long long *>(w + kMgWordReady); c.step = h.step; c.err = reinterpret_cast<int *>(w + kMgWordErr);
    c.grid_a = a.wg_long + a.wg_med + a.wg_short;
    c.grid_b = other ? b.wg_long + b.wg_med + b.wg_short : 0;
    c.n_poll = std::min(c.grid_b, std::max(1, h.max_pollers));
    c.mark = static_cast<const unsigned char *>(h.mark); c.mark_members = static_cast<const unsigned *>(h.mark_members);
    c.n_marked = other ? h.n_marked : 0; c.n_mark_shards = h.n_mark_shards;
    c.blk_order = static_cast<const int *>(h.blk_order);
    c.sleep = std::max(1, h.poll_sleep); c.timeout = h.timeout_ticks;
    const int grid = c.grid_a + c.n_poll;
    if (grid <= 0) { set_error("empty step"); return DASP_ERR_STATE; }
    hipStream_t s = static_cast<hipStream_t>(stream);
    // the own-column plan decides the cache policy of the streamed tiles (the other-column plan is a few per cent of the bytes)
    if (own.dev->nt) hipLaunchKernelGGL((dasp_mg_step_kernel<true>), dim3(grid), dim3(256), 0, s, a, b, c);
    else hipLaunchKernelGGL((dasp_mg_step_kernel<false>), dim3(grid), dim3(256), 0, s, a, b, c);
    HIP_TRY(hipGetLastError());
    return DASP_OK;
}

// workgroups of the step kernel one CU holds at a time (multigpu.cpp keeps one of them free of waiting workgroups)
int mg_step_resident_per_cu()
{
    int a = 0, b = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&a, reinterpret_cast<const void *>(&dasp_mg_step_kernel<true>), 256, 0) != hipSuccess) a = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&b, reinterpret_cast<const void *>(&dasp_mg_step_kernel<false>), 256, 0) != hipSuccess) b = 0;
    (void)hipGetLastError();
    return std::min(a, b);
}

int launch_mg_wait(const void *word, unsigned long long need, long long timeout_ticks, void *err, void *stream)
{
    hipLaunchKernelGGL(dasp_mg_wait_kernel, dim3(1), dim3(64), 0, static_cast<hipStream_t>(stream), static_cast<const unsigned long long *>(word), need,
                       timeout_ticks, static_cast<int *>(err));
    HIP_TRY(hipGetLastError());
    return DASP_OK;
}

int launch_mg_flag(void *word, unsigned long long value, void *stream)
{
    hipLaunchKernelGGL(dasp_mg_flag_kernel, dim3(1), dim3(64), 0, static_cast<hipStream_t>(stream), static_cast<unsigned long long *>(word), value);
    HIP_TRY(hipGetLastError());
    return DASP_OK;
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

// spmv_device.hpp -- the device side of the DASP SpMV on gfx950 (CDNA4, wave64): everything a kernel of this library is made of.
// Included by kernels.hip (the SpMV kernels proper) and mgstep.hip (the multi-GPU step kernels, which run the same bodies).
//
// Medium / long rows use the DASP diagonal trick on the CDNA4 matrix cores: a chunk of 16 rows x K columns is fed as A = values,
// B = x[column ids] with the same element index on both operands, so D[i][i] accumulates row i's dot product (reference: m8n8k4 PTX
// MMA, src/utils.h:102-115, dasp_f64.h:77-484; here v_mfma_f64_16x16x4_f64 / v_mfma_f32_16x16x16_f16).
// Lane maps (pinned on the device by dasp_selftest_mfma):
//   f64 16x16x4 : lane l holds A[l&15][l>>4], B[l>>4][l&15]; D reg r = D[(l>>4)+4r][l&15]
//   f16 16x16x16: lane l holds A[l&15][4(l>>4)+j], B[4(l>>4)+j][l&15], j<4; D reg r = D[4(l>>4)+r][l&15]
// Short rows (1..4 nonzeros) are uniform-length slabs: a lane owns whole rows, so the segmented dot product needs no cross-lane step;
// cross-lane sums (long rows, stage 2) use DPP row rotations + readlane.
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>

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

// The kernels' arguments: the plan's DevArgs lives on the device (DevicePlan::dargs) and is read word by word through the constant address space --
// scalar loads the compiler sinks to their uses and drops where a field is unused, exactly as it treats a by-value kernarg block -- and the four
// per-call values travel in the kernarg segment (device.hpp: ~0.7 us per small launch against the 528-byte by-value block).
#ifdef DASP_STAMPS
struct CallArgs { const DevArgs *plan; const void *x; void *y; int acc, ywt, stamp_launch; };
#else
struct CallArgs { const DevArgs *plan; const void *x; void *y; int acc, ywt; };
#endif
// a pointer field of the device-resident block, loaded AS a global pointer (address space 1) through the constant address space: a pointer that arrives in
// the kernarg segment is known to be global, one loaded from memory as a generic pointer is flat (flat_load instead of global_load, and no scalar
// loads through it -- the plain f64 kernels fell from 302 to 19 s_load that way)
template <class U> __device__ __forceinline__ U *ldp(U *const *field)
{
    typedef __attribute__((address_space(1))) U *gptr;
    return (U *)*reinterpret_cast<const __attribute__((address_space(4))) gptr *>(reinterpret_cast<uintptr_t>(field));
}
__device__ __forceinline__ DevArgs load_args(const CallArgs &c)
{
    static_assert(sizeof(DevArgs) % 4 == 0, "DevArgs is a whole number of words");
    int w[sizeof(DevArgs) / 4];
#pragma unroll
    for (int i = 0; i < (int)(sizeof(DevArgs) / 4); ++i) w[i] = tab<true>(reinterpret_cast<const int *>(c.plan), i);
    DevArgs a;
    __builtin_memcpy(&a, w, sizeof a);
#define DASP_G(f) a.f = ldp(&c.plan->f)
    DASP_G(long_val); DASP_G(long_cid); DASP_G(long_cid16); DASP_G(long_base); DASP_G(piece_c16); DASP_G(piece_ptr); DASP_G(piece_dst); DASP_G(partial); DASP_G(multi_ptr); DASP_G(multi_dst);
    DASP_G(med_ptr); DASP_G(med_val); DASP_G(med_cid); DASP_G(med_cid16); DASP_G(med_base); DASP_G(med_cid8); DASP_G(med_c8ptr);
    DASP_G(irr_ptr); DASP_G(med_nt); DASP_G(irr_val); DASP_G(irr_cid); DASP_G(med_dst); DASP_G(win_cmin); DASP_G(win_len);
    DASP_G(short_val); DASP_G(short_cid); DASP_G(groups); DASP_G(order);
    DASP_G(rt_val); DASP_G(rt_cid); DASP_G(rt_ptr); DASP_G(rt_start); DASP_G(rt_mask);
#undef DASP_G
    a.x = c.x; a.y = c.y; a.acc = c.acc; a.ywt = c.ywt;
    return a;
}

// the arrays INSIDE DevArgs are only ever read at compile-time indices: one run-time index, and the compiler keeps the array (or the whole block) in scratch and reaches
// it with flat loads -- which is also what a `#pragma unroll` loop turns into when a kernel grows and the unroller gives up (r5: the 36 bytes of xcd_blk in the row-tile
// kernel).  Template recursion cannot be "not unrolled".
// last_le: the last index i in [1, N) with w >= arr[i] (0 if none), and arr[that index] (0 for index 0)
template <int I, int N> struct LastLE {
    static __device__ __forceinline__ void run(const int (&arr)[N], int w, int &idx, int &val)
    {
        if constexpr (I < N) { const bool in = w >= arr[I]; idx = in ? I : idx; val = in ? arr[I] : val; LastLE<I + 1, N>::run(arr, w, idx, val); }
    }
};
// pick: arr[k], arr[k + 1] for a run-time k in [0, N - 1)
template <int I, int N> struct PickPair {
    static __device__ __forceinline__ void run(const int (&arr)[N], int k, int &lo, int &hi)
    {
        if constexpr (I + 1 < N) { lo = k == I ? arr[I] : lo; hi = k == I ? arr[I + 1] : hi; PickPair<I + 1, N>::run(arr, k, lo, hi); }
    }
};

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
    static constexpr int CHUNK = 64, SHORT_ROWS = 128, BATCH = kMedBatch64, SHOT = kMedShot64, LONG_SHOT = kMedShot64;
};
template <> struct Tr<_Float16> {
    using acc_t = f32x4; using part_t = float;
    static constexpr int CHUNK = 256, SHORT_ROWS = 256, BATCH = kMedBatch16, SHOT = kMedShot16, LONG_SHOT = 2;      // (long pieces of up to 4 chunks in one shot: rows of 300 0.524 -> 0.458)
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
// (the window's copy of x is held through an explicitly-LDS pointer and the global x of the hybrid through an explicitly-global one: as two generic pointers the optimiser
// folds XHyb's two branches into ONE flat load through a select of the pointers -- every staged gather then goes through the flat path, and with some register budgets the
// gfx950 backend of ROCm 7.2 fails on it outright ("Illegal instruction detected ... V_CMP_NE_U32_e32 0, $src_shared_base"); pointers of two address spaces cannot be merged)
template <class T> using lds_cptr = const __attribute__((address_space(3))) T *;
template <class T> using glb_cptr = const __attribute__((address_space(1))) T *;
template <class T>
struct XLds {
    lds_cptr<T> xw; int cmin;
    __device__ __forceinline__ T at(int c) const { return xw[c < 0 ? 0 : c - cmin]; }
};
// hybrid window: the densest span of the window's columns is in LDS, everything else is gathered from global memory.
// The two loads sit in divergent branches on purpose: a lane whose column is staged issues no global load.
template <class T>
struct XHyb {
    lds_cptr<T> xw; glb_cptr<T> xg; int cmin; unsigned len;
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

// a long piece: nfull whole chunks, then (tail > 0) its last, partial chunk as one more step of the SAME stream (r5: issued beside the whole chunks instead of behind
// their gathers -- a row of 300 is four chunks and a tail, and a wave's life was two dependent rounds of stream + gather instead of one: tools/category_sweep.py).
// Rows are padded to kLongAlign, so a lane's group of the tail is all-in or all-out; lanes beyond it re-read the piece's first group (in bounds) and gather x[0] times 0.
template <class T, bool NT, bool N16 = false>
struct PieceSrc {
    static constexpr bool kPairs = false, kQuadIds = false;
    static constexpr int VPL = Tr<T>::CHUNK / kWave;
    const T *val; const int *cid; size_t e0; int lane, nfull, tail;
    // N16 (r6, plan.hpp long_cid16): a NARROW piece reads its ids as u16 offsets -- 2 instead of 4 bytes per element -- from its chunks' base columns, a scalar load per
    // step; 0xFFFF = pad.  The same elements in the same lanes as the 32-bit form, so the same bits.  base = the piece's first entry of long_base.  An instantiation of its
    // own, chosen per piece: with a run-time flag inside ONE instantiation the wide pieces pay for the tests too (f16 rows of 300: 131 -> 144 us)
    const unsigned short *cid16 = nullptr; const int *base = nullptr;
    template <bool PAIRED_OK = true> __device__ __forceinline__ void load(Frag<T> &f, int i) const
    {
        if constexpr (N16) {
            const size_t at = i < nfull || VPL * lane < tail ? e0 + (size_t)i * Tr<T>::CHUNK + (size_t)(VPL * lane) : e0;
            if constexpr (VPL == 1) { f.a = ldg<NT>(val + at); f.c = (int)ldg<NT>(cid16 + at); }
            else {
                f.a = ldg<NT>(reinterpret_cast<const f16x4 *>(val + at));
                const i32x2 o = ldg<NT>(reinterpret_cast<const i32x2 *>(cid16 + at));      // raw u16 offsets, two per dword; unpacked and rebased in gather()
                f.c[0] = o[0]; f.c[1] = o[1];
            }
        } else {
            if (i < nfull) frag_load<NT>(f, val, cid, e0 + (size_t)i * Tr<T>::CHUNK, lane);
            else frag_load_at<NT>(f, val, cid, VPL * lane < tail ? e0 + (size_t)i * Tr<T>::CHUNK + (size_t)(VPL * lane) : e0);
        }
    }
    template <bool QUAD = false, class XV> __device__ __forceinline__ void gather(Frag<T> &f, int i, const XV &x) const
    {
        if constexpr (N16) {
            const int b = tab<true>(base, i);                   // wave-uniform: one scalar load per step
            if constexpr (VPL == 1) { const unsigned o = (unsigned)f.c; f.c = o == kLongPad16 ? -1 : b + (int)o; }
            else {
                const unsigned lo = (unsigned)f.c[0], hi = (unsigned)f.c[1];
                const unsigned o[4] = {lo & 0xFFFFu, lo >> 16, hi & 0xFFFFu, hi >> 16};
#pragma unroll
                for (int q = 0; q < 4; ++q) f.c[q] = o[q] == kLongPad16 ? -1 : b + (int)o[q];
            }
        }
        if (i >= nfull) {
            const bool ok = VPL * lane < tail;
            if constexpr (VPL == 1) { f.c = ok ? f.c : -1; f.a = ok ? f.a : 0.0; }      // (the value too: the re-read element may be inf / NaN)
            else {
#pragma unroll
                for (int j = 0; j < 4; ++j) { f.c[j] = ok ? f.c[j] : -1; f.a[j] = ok ? f.a[j] : (_Float16)0; }
            }
        }
        frag_gather(f, x);
    }
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
                if constexpr (kQuadIds) {
                    // (load2 of this instantiation serves the ONE-SHOT blocks only: the pipeline uses load4.)  A narrow pair (i < n8, r4) keeps its
                    // ids as one 16-bit word per lane, [pair][lane][2 bytes]; the pad 0xFF becomes the wide pad, so that gather() needs no second form
                    if (i < n8) {              // wave-uniform
                        const unsigned r = ldg<NT>(reinterpret_cast<const unsigned short *>(c8 + (size_t)i * CH) + reg.lane);
                        const unsigned b0 = r & 0xFFu, b1 = r >> 8;
                        f0.c = (int)(b0 == 0xFFu ? 0xFFFFu : b0); f1.c = (int)(b1 == 0xFFu ? 0xFFFFu : b1);
                        return;
                    }
                }
                const unsigned r = ldg<NT>(reinterpret_cast<const unsigned *>((kQuadIds ? w16 : cid16) + at));      // raw offsets; rebased in gather()
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
    }
    else if (a.acc) *y = (T)((P)*y + v);
    // written through (sc0 sc1) where the plan streams from HBM (DevArgs::ywt, f64): the y lines then leave the XCD's L2 at once instead of as
    // dirty evictions under the read stream.  Same-device A/B on three boxes (profiles/r04_placement.md): HV15R 0.4750 -> 0.4590, 0.4597 ->
    // 0.4472, Queen_4147 0.6118 -> 0.5896, nlpkkt160 0.4799 -> 0.4750 on slow placements, +-0.2 % on fast ones.  volatile, not an atomic
    // store: see YS == 1 above (the wave ends behind its store, so the wait a volatile access brings costs nothing)
    else if (a.ywt) *(volatile T *)y = (T)v;
    else *y = (T)v;
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
// WU / WS (windowed f64 only; 0 = the defaults of Tr<T>): pipeline batch and longest one-shot unit of the build held to 64 registers -- with 4 + 4 fragments in flight it
// spilled 12-29 VGPRs; windowed plans store no pairs, so their layout does not depend on the batch
template <class T, bool NT, bool C16, int YM, bool C8 = false, int YS = 0, bool REL = false, int WU = 0, int WS = 0, class XV>
__device__ __forceinline__ void medium_block(const DevArgs &a, int b, int lane, const XV &x)
{
    using acc_t = typename Tr<T>::acc_t;
    constexpr int CH = Tr<T>::CHUNK;
    const T *val = static_cast<const T *>(a.med_val);
    // tables through the CONSTANT address space (tab<>) wherever the compiler could not prove them unclobbered: the step kernels (YS != 0) and the windowed
    // kernels, whose blocks sit behind the barrier of the x copy (r4: cop20k_A x16, 1695 windows, 124.4 -> 111.1 us; the 212-window size is unchanged)
    constexpr bool KT = true;
    const int c0 = tab<KT>(a.med_ptr, b), c1 = tab<KT>(a.med_ptr, b + 1);
    acc_t acc = {0, 0, 0, 0};
    // the block's first row is its longest (rows are sorted), so its tail length bounds the number of tail steps
    const int row = lane & 15, kq = lane >> 4;
    const int r = b * kMedRows + row;
    int t0 = 0, t1 = 0;
    if (r < a.row_block) { t0 = a.irr_ptr[r]; t1 = a.irr_ptr[r + 1]; }
    constexpr int TK = sizeof(T) == 8 ? 4 : 16;                      // tail entries of one row per MFMA step
    // (r6: from the per-block table, a scalar load beside med_ptr's -- the wave does not wait for the vector load of irr_ptr before it issues its tiles' loads; the multi-GPU
    // step kernels, YS != 0, keep the form they were frozen with)
    const int nt = YS == 0 ? tab<true>(a.med_nt, b) : (__builtin_amdgcn_readfirstlane(t1 - t0) + TK - 1) / TK;
    BlockSrc<T, NT, C16, YM != 2, C8, KT, REL> src;
    if constexpr (REL) src.relb = x.cmin; else src.relb = 0;
    src.reg.val = val; src.reg.cid = a.med_cid; src.reg.e0 = (size_t)c0 * CH; src.reg.lane = lane;
    src.nc = c1 - c0; src.npair = med_npair(c1 - c0, nt, (int)sizeof(T), YM == 2 ? 0 : a.pair_mode); src.cid16 = a.med_cid16; src.base = a.med_base; src.c0 = c0;
    src.c8 = a.med_cid8; src.w16 = a.med_cid16; src.n8 = 0;
    if constexpr (C8 && C16 && sizeof(T) == 8 && YM != 2) {
        const int q0 = tab<true>(a.med_c8ptr, b), q1 = tab<true>(a.med_c8ptr, b + 1);
        src.n8 = q1 - q0; src.c8 = a.med_cid8 + (size_t)q0 * CH; src.w16 = a.med_cid16 - (size_t)q1 * CH;      // e16 - (e0 + n8 CH) = -(q0 + n8) CH
    }
    src.ival = static_cast<const T *>(a.irr_val); src.icid = a.irr_cid; src.t0 = t0; src.t1 = t1; src.kq = kq;
    // (the windowed f16 kernels, held to 64 registers, keep blocks of up to 2 steps in one shot: 4 spill there; windowed plans store no pairs, so their layout does not depend on it)
    run_stream<T, (WU > 0 && sizeof(T) == 8) ? WU : Tr<T>::BATCH, (YM == 2 && sizeof(T) == 2) ? 2 : ((WS > 0 && sizeof(T) == 8) ? WS : Tr<T>::SHOT)>(acc, src, src.nc + nt, x);

    typename Tr<T>::part_t d;
    if (diag_of(acc, lane, d) && r < a.row_block) {
        const int slot = a.row_long + r;                 // row_long here = slot of the first MFMA medium row (Plan::med_slot0)
        const int yi = YM == 2 ? a.med_dst[r] : (a.order ? a.order[slot] : slot);
        put_y<T, YS>(a, yi, d);
    }
}

// ---- long: one wave = one piece (<= long_piece elements) of one long row (reference: dasp_f64.h:90-144)
// TAIL_IN: the partial last chunk rides in the stream (PieceSrc) -- the plain kernels; the windowed and the multi-GPU step kernels, which sit at their register caps,
// keep it as a step of its own behind the stream
// L16: the build reads the 16-bit ids of narrow pieces (plan.hpp long_cid16)
template <class T, bool NT, int YS = 0, bool TAIL_IN = false, bool L16 = false>
__device__ __forceinline__ void long_piece(const DevArgs &a, int p, int lane)
{
    using acc_t = typename Tr<T>::acc_t;
    using part_t = typename Tr<T>::part_t;
    constexpr int CH = Tr<T>::CHUNK;
    constexpr int VPL = CH / kWave;          // values per lane per MFMA: 1 (f64) / 4 (f16)
    const XGlobal<T> x{static_cast<const T *>(a.x)};
    const T *val = static_cast<const T *>(a.long_val);
    const int p0 = tab<true>(a.piece_ptr, p), p1 = tab<true>(a.piece_ptr, p + 1);
    acc_t acc = {0, 0, 0, 0};
    const int nfull = (p1 - p0) / CH, tail = (p1 - p0) - nfull * CH;
    // narrow piece: 16-bit ids (L16: the plain kernels without one-byte ids and the merged-panel kernels -- the one-byte-id and 7-wave f64 builds, the windowed and the
    // multi-GPU step kernels sit at their register ceilings and read the 32-bit ids every piece keeps; the r6 attempt cost the one-byte-id build a 36-byte frame)
    bool narrow = false;
    if constexpr (L16) narrow = tab<true>(a.piece_c16, 2 * p + 1) != 0;
    if (L16 && narrow) {
        const PieceSrc<T, NT, true> src{val, a.long_cid, (size_t)p0, lane, nfull, tail, a.long_cid16, a.long_base + tab<true>(a.piece_c16, 2 * p)};
        run_stream<T, Tr<T>::BATCH, Tr<T>::LONG_SHOT>(acc, src, nfull + (tail > 0 ? 1 : 0), x);
    } else if (TAIL_IN && tail > 0) {       // (pieces of whole chunks -- every piece but the last of a row cut in pieces -- keep the stream without the per-step test: rmat_2M f64 +1.3 % with it)
        const PieceSrc<T, NT> src{val, a.long_cid, (size_t)p0, lane, nfull, tail};
        run_stream<T, Tr<T>::BATCH, Tr<T>::LONG_SHOT>(acc, src, nfull + 1, x);
    } else {
        const int full = p0 + nfull * CH;
        ChunkSrc<T, NT> src{val, a.long_cid, (size_t)p0, lane};
        run_stream<T, Tr<T>::BATCH, Tr<T>::LONG_SHOT>(acc, src, nfull, x);
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
    }
    part_t d;
    const bool on_diag = diag_of(acc, lane, d);
    const part_t total = wave_sum(on_diag ? d : (part_t)0);
    if (lane == 0) {
        const int dst = tab<true>(a.piece_dst, p);
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

// ---- short rows, wave-segmented (ShortGroup::seg; plan.hpp short_elem_index): one wave = one tile of 64 ELEMENTS, one nonzero per lane; every
// 16-lane DPP row holds 16 / L whole rows back to back (L = 3: five rows, lane 15 idle).  The products of a row are summed towards its first
// lane with two DPP row shifts -- lane i adds lane i + 1, then the head adds lane i + 2: (p0 + p1) + (p2 + p3) -- and the head lanes store y:
// the north_star's "wavefront-segmented dot product with DPP reductions" (reference: the four short-row branches of dasp_spmv2,
// dasp_f64.h:281-483, which fill 8x4 MMA tiles with pairs of short rows instead).
template <class T, int L, bool NT, int YS = 0>
__device__ __forceinline__ void short_rows_seg(const DevArgs &a, const ShortDev &g, int local_tile, int lane)
{
    constexpr int PER16 = 16 / L, RPW = 4 * PER16;
    using part_t = typename Tr<T>::part_t;
    const T *x = static_cast<const T *>(a.x);
    const int sub = lane & 15, rloc = sub / L, k = sub - rloc * L;
    const int t = local_tile * RPW + (lane >> 4) * PER16 + rloc;          // the row of the group this lane works for
    const size_t e = (size_t)g.elem_off + (size_t)local_tile * kWave + (size_t)lane;
    const T av = ldg<NT>(static_cast<const T *>(a.short_val) + e);
    const int c = ldg<NT>(a.short_cid + e);
    const T xv = x[c < 0 ? 0 : c];                                        // pads (idle lanes, the tile behind the last row): clamped gather, value dropped
    part_t p = c < 0 ? (part_t)0 : (part_t)av * (part_t)xv;
    if constexpr (L >= 2) {
        part_t q;
        if constexpr (sizeof(part_t) == 8) q = dpp_mov_f64<0x101>(p); else q = dpp_mov_f32<0x101>(p);      // row_shl:1 -- lane i reads lane i + 1
        if (k + 1 < L) p += q;
    }
    if constexpr (L >= 3) {
        part_t q;
        if constexpr (sizeof(part_t) == 8) q = dpp_mov_f64<0x102>(p); else q = dpp_mov_f32<0x102>(p);      // row_shl:2
        if (k == 0) p += q;
    }
    if (k == 0 && rloc < PER16 && t < g.count) {
        const int slot = slot_of(g.map, t);
        const int yi = a.order ? a.order[slot] : slot;
        put_y<T, YS>(a, yi, p);
    }
}

// TPW consecutive tiles of ONE wave-segmented group in one wave (r5): every load of the TPW tiles is issued before the first gather, every gather before the first
// DPP step.  nv (wave-uniform) = how many of the tiles exist; the others repeat the first tile's addresses and store nothing.
template <class T, int L, bool NT, int TPW>
__device__ __forceinline__ void short_rows_seg_multi(const DevArgs &a, const ShortDev &g, int local0, int nv, int lane)
{
    constexpr int PER16 = 16 / L, RPW = 4 * PER16;
    using part_t = typename Tr<T>::part_t;
    const T *x = static_cast<const T *>(a.x);
    const int sub = lane & 15, rloc = sub / L, k = sub - rloc * L;
    T av[TPW]; int c[TPW]; T xv[TPW];
#pragma unroll
    for (int u = 0; u < TPW; ++u) {
        const size_t e = (size_t)g.elem_off + (size_t)(local0 + (u < nv ? u : 0)) * kWave + (size_t)lane;
        av[u] = ldg<NT>(static_cast<const T *>(a.short_val) + e);
        c[u] = ldg<NT>(a.short_cid + e);
    }
#pragma unroll
    for (int u = 0; u < TPW; ++u) xv[u] = x[c[u] < 0 ? 0 : c[u]];
#pragma unroll
    for (int u = 0; u < TPW; ++u) {
        part_t p = c[u] < 0 ? (part_t)0 : (part_t)av[u] * (part_t)xv[u];
        if constexpr (L >= 2) {
            part_t q;
            if constexpr (sizeof(part_t) == 8) q = dpp_mov_f64<0x101>(p); else q = dpp_mov_f32<0x101>(p);
            if (k + 1 < L) p += q;
        }
        if constexpr (L >= 3) {
            part_t q;
            if constexpr (sizeof(part_t) == 8) q = dpp_mov_f64<0x102>(p); else q = dpp_mov_f32<0x102>(p);
            if (k == 0) p += q;
        }
        const int t = (local0 + u) * RPW + (lane >> 4) * PER16 + rloc;
        if (u < nv && k == 0 && rloc < PER16 && t < g.count) {
            const int slot = slot_of(g.map, t);
            const int yi = a.order ? a.order[slot] : slot;
            put_y<T>(a, yi, p);
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

// SEG: the wave-segmented short-row path is compiled in (not in the windowed kernels: they are held to 64 registers, and windowed plans never
// store segmented groups)
template <class T, bool NT, int YS = 0, bool SEG = true>
__device__ __forceinline__ void short_tile(const DevArgs &a, int tile, int lane)
{
    int gi = 0, t0_unused = 0;
    LastLE<1, kNumShortGroups>::run(a.grp_tile0, tile, gi, t0_unused);     // scalar compares
    ShortDev g;
    if constexpr (YS != 0) {      // word by word through the constant address space (see tab)
        static_assert(sizeof(ShortDev) % 4 == 0, "ShortDev is a whole number of words");
        int w[sizeof(ShortDev) / 4];
#pragma unroll
        for (int i = 0; i < (int)(sizeof(ShortDev) / 4); ++i) w[i] = tab<true>(reinterpret_cast<const int *>(a.groups + gi), i);
        __builtin_memcpy(&g, w, sizeof g);
    } else g = a.groups[gi];
    const int local = tile - g.tile0;
    if constexpr (SEG) {
        if (g.seg) {
            switch (g.len) {
                case 1: short_rows_seg<T, 1, NT, YS>(a, g, local, lane); break;
                case 2: short_rows_seg<T, 2, NT, YS>(a, g, local, lane); break;
                case 3: short_rows_seg<T, 3, NT, YS>(a, g, local, lane); break;
                default: short_rows_seg<T, 4, NT, YS>(a, g, local, lane); break;
            }
            return;
        }
    }
    switch (g.len) {
        // empty rows: y = 0 -- nothing to do for y += 0, nor for a column panel, whose partial-result buffer is zeroed ONCE at upload and never
        // written at the slots of the rows that are empty in that panel (a third to a half of all (row, panel) pairs of the graph stand-ins)
        case 0: if constexpr (YS != 2) { if (!a.skip0 && !a.acc) short_rows<T, 0, NT, YS>(a, g, local, lane); } break;
        case 1: short_rows<T, 1, NT, YS>(a, g, local, lane); break;
        case 2: short_rows<T, 2, NT, YS>(a, g, local, lane); break;
        case 3: short_rows<T, 3, NT, YS>(a, g, local, lane); break;
        case 4: short_rows<T, 4, NT, YS>(a, g, local, lane); break;
        default: slab_rows<T, NT, YS>(a, g, local, lane); break;
    }
}

// wave w of the short range of a plan with wave-segmented groups (DevArgs::grp_wave0): TPW consecutive tiles of a segmented group, one tile of any other group -- a
// wave never straddles two groups, so there is no loop here (a store followed by another tile's loads in one function costs the f64 kernels their scalar table loads:
// tests/test_isa_guard.py)
template <class T, bool NT, int TPW>
__device__ __forceinline__ void short_waves(const DevArgs &a, int w, int lane)
{
    int gi = 0, w0 = 0;
    LastLE<1, kNumShortGroups>::run(a.grp_wave0, w, gi, w0);
    const ShortDev g = a.groups[gi];
    const int lw = w - w0;
    if (!g.seg) { short_tile<T, NT, 0, false>(a, g.tile0 + lw, lane); return; }
    const int local0 = lw * TPW, nv = min(TPW, g.tiles - local0);
    switch (g.len) {
        case 1: short_rows_seg_multi<T, 1, NT, TPW>(a, g, local0, nv, lane); break;
        case 2: short_rows_seg_multi<T, 2, NT, TPW>(a, g, local0, nv, lane); break;
        case 3: short_rows_seg_multi<T, 3, NT, TPW>(a, g, local0, nv, lane); break;
        default: short_rows_seg_multi<T, 4, NT, TPW>(a, g, local0, nv, lane); break;
    }
}

// launch bounds: the windowed kernel is held to 64 registers so that two 1024-thread window workgroups share a CU
// (A/B: 12.9 vs 15.0 us on cop20k_A); blocks are dealt to workgroups in the default round-robin order (length-sorted
// blocks in XCD-contiguous ranges put all the long ones on one XCD: DESIGN.md 4.4)
//
// ---- row tile of a column panel (Plan::rt_*, no reference counterpart): 64 consecutive positions of the parent's output order, the rows of
// at most rt_max nonzeros among them.  The wave streams the tile's elements (CSR order, one per lane and step), parks the products in its
// LDS slice, then lane r adds row r's products in their CSR order and the wave stores one complete line (f16) / two lines (f64) of the
// panel's partial result -- the positions of longer rows (mask bit 0) are left to the panel's blocks.  prod: this wave's 64 * rt_max sums.
template <class T, bool NT>
__device__ __forceinline__ void row_tile(const DevArgs &a, int t, int lane, typename Tr<T>::part_t *prod)
{
    using part_t = typename Tr<T>::part_t;
    const T *x = static_cast<const T *>(a.x), *val = static_cast<const T *>(a.rt_val);
    const int e0 = a.rt_ptr[t], n = a.rt_ptr[t + 1] - e0;
    const int s = a.rt_start[(size_t)t * kRowTile + lane];
#pragma unroll 4
    for (int i = lane; i < n; i += kWave) prod[i] = (part_t)ldg<NT>(val + e0 + i) * (part_t)x[ldg<NT>(a.rt_cid + e0 + i)];
    // this wave's LDS writes before its LDS reads (another lane's): one wave's LDS operations execute in issue order, and the compiler keeps a store before a
    // load of the same array.  NO fence and NO __builtin_amdgcn_wave_barrier() here: either makes the compiler read the row tables of the panel's blocks with
    // vector instead of scalar loads in the f64 kernel (the cliff of DESIGN_MULTIGPU.md 5.1: 302 -> 119 s_load instructions; HV15R-unstructured in two forced panels
    // 0.555 -> 0.884 ms with 0.05 % of its nonzeros in tiles)
    int e = __shfl_down(s, 1);
    if (lane == kWave - 1) e = n;
    part_t sum = 0;
    for (int j = s; j < e; ++j) sum += prod[j];
    if ((a.rt_mask[t] >> lane) & 1) put_y<T>(a, t * kRowTile + lane, sum);
}

// ---- per-phase time stamps of a workgroup (tools/stamp_probe.py; the DASP_STAMPS build of tools/build_variant.sh only -- in the product every member is empty and the
// calls vanish).  mark(i) waits for everything the wave has in flight, then reads the shader clock: phase i ends where all loads issued before it have returned.
// finish() adds a workgroup barrier, takes a ticket (completion order; ticket / grid = the launch) and writes one 16-word record.
#ifdef DASP_STAMPS
struct StampBuf { unsigned long long *rec; unsigned cap; };      // cap = records (one per wave, 12 words each)
static __device__ StampBuf g_stamps;
struct Stamps {
    unsigned long long wall0, clk[4];
    int kind, launch;
    __device__ __forceinline__ void begin(int launch_no) { wall0 = __builtin_amdgcn_s_memrealtime(); clk[0] = __builtin_readcyclecounter(); clk[1] = clk[2] = clk[3] = clk[0]; kind = 0; launch = launch_no; }
    __device__ __forceinline__ void mark(int i) { asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory"); clk[i] = __builtin_readcyclecounter(); }
    // no barrier, no atomic (3000 workgroups taking tickets from one word serialise at the memory side: the stamped launch took 3x as long): the record's place follows from
    // the launch number the host passes, the workgroup and the wave
    __device__ __forceinline__ void finish(int nwaves)
    {
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        const unsigned long long e = __builtin_readcyclecounter();
        const unsigned long long wall1 = __builtin_amdgcn_s_memrealtime();
        const size_t slot = ((size_t)launch * gridDim.x + blockIdx.x) * (size_t)nwaves + (threadIdx.x >> 6);
        if ((threadIdx.x & 63) == 0 && g_stamps.rec && launch >= 0 && slot < g_stamps.cap) {
            unsigned long long *r = g_stamps.rec + slot * 12;
            unsigned xcc, hw;
            asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
            asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
            r[0] = blockIdx.x | ((unsigned long long)(threadIdx.x >> 6) << 32) | ((unsigned long long)nwaves << 40) | ((unsigned long long)kind << 48) | (1ull << 63);
            r[1] = xcc | ((unsigned long long)hw << 32); r[2] = gridDim.x; r[3] = wall0; r[4] = wall1;
            r[5] = clk[0]; r[6] = clk[1]; r[7] = clk[2]; r[8] = clk[3]; r[9] = e; r[10] = (unsigned long long)launch; r[11] = 0;
        }
    }
};
#else
struct Stamps {
    int kind;
    __device__ __forceinline__ void begin(int) {}
    __device__ __forceinline__ void mark(int) {}
};
#endif

#ifndef DASP_MIN_WAVES
#define DASP_MIN_WAVES 1
#endif
constexpr int kMinWavesPlain = DASP_MIN_WAVES, kMinWavesWin = 8;
// WIN: windowed mode.  A medium workgroup owns one window of row_window rows (blocks_per_win blocks, strided over its
// 4 waves); if the window's x span fits, it is copied once into LDS with coalesced 16-byte loads and every gather of
// the window reads LDS; otherwise that workgroup gathers from global memory like the non-windowed kernel.
// the 64-register build's pipeline batch / longest one-shot unit (r6, same box, cop20k_A x4 / x16 / x64 f64): (4, 8) -- the other kernels' values, 12-29 spilled VGPRs --
// 26.75 / 114.1 / 438.0 us; (2, 8) 26.2 / 104.7 / 416.9; (3, 6) 26.9 / 103.1 / 412.9: no scratch from (3, x) down
#ifndef DASP_WIN64_U
#define DASP_WIN64_U 3
#define DASP_WIN64_S 6
#endif
// W64: the windowed build held to 64 registers (two window workgroups per CU)
template <class T, bool NT, bool C16, bool WIN, bool C8, bool RT = false, bool W64 = false, bool L16 = false>
__device__ __forceinline__ void spmv_body(const DevArgs &a, char *lds_raw, int wg_index, Stamps &st)
{
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    // blockIdx.x, or the workgroup's index inside its panel's range of a merged launch (dasp_spmv_panels_kernel) -- rotated by wg_rot (plain plans only: 0 elsewhere)
    int wg = wg_index;
    if constexpr (sizeof(T) == 2 && !WIN && !RT) {      // (f16 builds only: the f64 one-byte-id build, at its 104 SGPRs, pays for it with a 36-byte frame)
        if (a.wg_rot) { const int grid = a.wg_long + a.wg_med + a.wg_short; wg += a.wg_rot; wg = wg >= grid ? wg - grid : wg; }
    }
    const int wpw = WIN ? a.wpw : kWavesPerWG;
    if (wg < a.wg_long) {
        const int p = wg * wpw + wave;
        if (p < a.n_pieces) long_piece<T, NT, 0, !WIN, L16 && !WIN>(a, p, lane);
    } else if (wg < a.wg_long + a.wg_med) {
        st.kind = 1;
        if constexpr (!WIN) {
            const int m = wg - a.wg_long;
            const XGlobal<T> x{static_cast<const T *>(a.x)};
            if (a.xcd_on) {
                // workgroups go to the XCDs round-robin, each XCD has an L2 of its own: with the blocks dealt round-robin too, every
                // XCD gathers from ALL of x (nlpkkt160: 8 x 67 MB of x through the L2s against 2.4 GB of matrix).  Rows of equal length
                // keep their row order in the sort, so a contiguous range of blocks is a contiguous part of the mesh: XCD k takes the
                // k-th eighth of the blocks (eighths of equal work) and touches an eighth of x plus the halo.
                // (constant indices + scalar selects instead of xcd_blk[k]: a run-time index into DevArgs would pin the whole block in scratch, load_args)
                const int k = m & 7;
                int lo = a.xcd_blk[0], hi = a.xcd_blk[1];
                PickPair<1, 9>::run(a.xcd_blk, k, lo, hi);
                const int b = lo + (m >> 3) * kWavesPerWG + wave;
                if (b < hi) medium_block<T, NT, C16, 0, C8>(a, b, lane, x);
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
            // one window per workgroup.  Workgroups are dealt to the 8 XCDs round-robin, so workgroup m of the range takes window (m % 8) * per_xcd + m / 8: every
            // XCD works on ONE contiguous eighth of the windows -- the same eighth in every launch, whose tiles and x (an eighth of x plus the band) can stay in
            // that XCD's 4 MiB L2 from one SpMV to the next when the matrix is small enough (cop20k_A: 3.5 MB per XCD)
            const int mw = wg - a.wg_long, per_xcd = a.wg_med >> 3;
            const int w = a.win_xcd ? (mw & 7) * per_xcd + (mw >> 3) : mw;
            if (w >= a.n_windows) return;
            const int len = a.win_len[w], cmin = a.win_cmin[w];
            const T *xg = static_cast<const T *>(a.x);
            // the workgroup's units: its blocks_per_win blocks, longest first (the window's rows are sorted by length), then the short tiles folded into it
            // (DevArgs::win_tiles: tiles w, w + n_windows, ...).  Wave k starts with unit k and then takes the next free one from a counter in LDS: the waves end
            // together whatever the blocks' lengths (r6, profiles/r06_small_matrix.md: dealt round-robin, a workgroup's first wave was done after 4.5 us, its last
            // after 6.0), and the short tiles -- 16 waves of 64-line gathers per CU when they had workgroups of their own, the last waves of the launch to exit --
            // are spread over all CUs as fillers
            __shared__ int next_unit_word;                // (static LDS beside the dynamic window: kWinLdsMax leaves room for it)
            int *next_unit = &next_unit_word;
            typedef __attribute__((address_space(3))) T lds_T;
            lds_T *xw = (lds_T *)lds_raw;                 // (lds_raw IS the kernel's dynamic LDS: the cast back folds away once spmv_body is inlined)
            if (threadIdx.x == 0) *next_unit = wpw;
            st.kind = 3; st.mark(1);
            if (len > 0) {
                constexpr int A = 16 / (int)sizeof(T);
                const i32x4 *src = reinterpret_cast<const i32x4 *>(xg + cmin);
                typedef __attribute__((address_space(3))) i32x4 lds_i32x4;
                lds_i32x4 *dst = (lds_i32x4 *)xw;
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
                st.mark(2);
            }
            __syncthreads();
            st.mark(3);
            const int n_units = a.blocks_per_win + a.win_tiles;
            // (one loop per gather source: the block is a different instantiation in each.  The short tiles have a loop of their own behind the blocks' -- in ONE loop with
            // the LDS-gathering blocks the f16 64-register build does not compile: "Illegal instruction detected ... V_CMP_NE_U32_e32 0, $src_shared_base", ROCm 7.2)
            constexpr int WU = W64 ? DASP_WIN64_U : 0, WS = W64 ? DASP_WIN64_S : 0;
            int q = wave;
#define DASP_NEXT_UNIT()                                                                                          \
            {                                                                                                     \
                int nq = 0;                                                                                       \
                if (lane == 0) nq = atomicAdd(next_unit, 1);                                                      \
                q = __builtin_amdgcn_readfirstlane(nq);                                                           \
            }
#define DASP_WINDOW_BLOCKS(BLOCK_CALL)                                                                            \
            while (q < a.blocks_per_win) {                                                                        \
                const int b = w * a.blocks_per_win + q;                                                           \
                if (b < a.n_blocks) { BLOCK_CALL; }                                                               \
                DASP_NEXT_UNIT()                                                                                  \
            }
            if (len > 0) {
                if (a.win_hybrid) {
                    const XHyb<T> x{xw, (glb_cptr<T>)xg, cmin, (unsigned)len};
                    DASP_WINDOW_BLOCKS((medium_block<T, NT, C16, 2, false, 0, false, WU, WS>(a, b, lane, x)))
                } else if (C16 && a.win_rel16) {
                    const XLds<T> x{xw, cmin};
#ifdef DASP_WIN_KEEPMOD     // experiment build (tools/build_variant.sh): all windows but every DASP_WIN_KEEPMOD-th read their tiles with plain loads (could they stay in the L2 from launch to launch?)
                    if ((w % DASP_WIN_KEEPMOD) != DASP_WIN_KEEPMOD - 1) { DASP_WINDOW_BLOCKS((medium_block<T, false, C16, 2, false, 0, C16, WU, WS>(a, b, lane, x))) }
                    else
#endif
                    DASP_WINDOW_BLOCKS((medium_block<T, NT, C16, 2, false, 0, C16, WU, WS>(a, b, lane, x)))
                } else {
                    const XLds<T> x{xw, cmin};
                    DASP_WINDOW_BLOCKS((medium_block<T, NT, C16, 2, false, 0, false, WU, WS>(a, b, lane, x)))
                }
            } else {
                const XGlobal<T> x{xg};
                DASP_WINDOW_BLOCKS((medium_block<T, NT, C16, 2, false, 0, false, WU, WS>(a, b, lane, x)))
            }
            while (q < n_units) {
                const int t = w + (q - a.blocks_per_win) * a.n_windows;
                if (t < a.n_short_tiles) short_tile<T, NT, 0, false>(a, t, lane);
                DASP_NEXT_UNIT()
            }
#undef DASP_WINDOW_BLOCKS
#undef DASP_NEXT_UNIT
        }
    } else if (!RT || wg < a.wg_long + a.wg_med + a.wg_short) {
        st.kind = 2;
        const int t = (wg - a.wg_long - a.wg_med) * wpw + wave;
        if constexpr (!WIN && sizeof(T) == 8) {      // (f64 only: the f16 kernels are held to 72 registers and never segment their short rows by themselves)
            if (a.short_tpw > 1) {      // plans with wave-segmented groups: kShortTpw tiles per wave (upload_plan)
                if (t < a.n_short_waves) short_waves<T, NT, kShortTpw>(a, t, lane);
                return;
            }
        }
        if (t < a.n_short_tiles) short_tile<T, NT, 0, !WIN>(a, t, lane);
    } else if constexpr (RT) {
        const int t = (wg - a.wg_long - a.wg_med - a.wg_short) * kWavesPerWG + wave;
        if (t < a.n_rt_tiles) row_tile<T, NT>(a, t, lane, reinterpret_cast<typename Tr<T>::part_t *>(lds_raw) + (size_t)wave * kRowTile * a.rt_max);
    }
}

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
        // (the multi-GPU step kernels, YS != 0, do without the wave-segmented short rows: with them they need 83-85 registers, one wave per SIMD less;
        // multigpu.cpp builds its plans with short_seg off)
        if (t < a.n_short_tiles) short_tile<T, NT, YS, YS == 0>(a, t, lane);
    }
}

}  // namespace dasp

// devpack.hip -- the O(nnz) half of plan creation on the GPU (dasp_plan_create_device).
// The reference's preprocessing is serial host code (src/dasp_f64.h:499-1157; f16 times it as "dasp_pre",
// dasp_f16.h:1444-1445).  Here, when the CSR already lives on the device, the host still takes every O(rows) decision
// from the row pointer alone (plan.cpp) and these kernels do the O(nnz) work in place: range check of the column ids,
// column spans of windows / chunks, and the copy of every nonzero into the packed arrays of the plan's arena --
// the same bytes plan.cpp's host packers produce (tests compare them bit for bit).
#include <hip/hip_runtime.h>
#include <hipcub/hipcub.hpp>

#include <memory>
#include <thread>

#include <algorithm>
#include <cstring>
#include <string>
#include <vector>

#include "device.hpp"

namespace dasp {

#define HIP_TRYP(expr)                                                                         \
    do {                                                                                       \
        hipError_t e_ = (expr);                                                                \
        if (e_ != hipSuccess) {                                                                \
            set_error(std::string(#expr) + ": " + hipGetErrorString(e_));                      \
            return e_ == hipErrorNoDevice ? DASP_ERR_NO_DEVICE : DASP_ERR_HIP;                 \
        }                                                                                      \
    } while (0)

namespace {

struct RemapDev {            // column remap of the row-partitioned layout (plan.cpp Remap), bounds on the device
    const int *bounds; int n_parts, stride;
    __device__ __forceinline__ int operator()(int c) const
    {
        if (n_parts <= 0) return c;
        int lo = 0, hi = n_parts;              // last g with bounds[g] <= c
        while (hi - lo > 1) { const int mid = (lo + hi) >> 1; if (bounds[mid] <= c) lo = mid; else hi = mid; }
        return lo * stride + (c - bounds[lo]);
    }
};

__device__ __forceinline__ int wave_min(int v) { for (int o = 32; o; o >>= 1) v = min(v, __shfl_xor(v, o)); return v; }
__device__ __forceinline__ int wave_max(int v) { for (int o = 32; o; o >>= 1) v = max(v, __shfl_xor(v, o)); return v; }

__global__ void k_validate(const int *ci, long long nnz, int ncol, int *bad)
{
    const long long stride = (long long)gridDim.x * blockDim.x;
    int any = 0;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < nnz; i += stride) any |= (unsigned)ci[i] >= (unsigned)ncol;
    if (any) atomicOr(bad, 1);
}

// one wave per window of R consecutive positions of ridW: min / max remapped column, nonzero count
__global__ void k_window_spans(const int *rp, const int *ci, const int *ridW, int nmed, int R, RemapDev remap, int *lo_out, int *hi_out,
                               long long *nnz_out, int n_windows)
{
    const int lane = threadIdx.x & 63, w = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (w >= n_windows) return;
    const int a0 = w * R, a1 = min(nmed, a0 + R);
    int lo = 2147483647, hi = -1; long long k = 0;
    for (int i = a0; i < a1; ++i) {
        const int r = ridW[i], b = rp[r], e = rp[r + 1];
        for (int j = b + lane; j < e; j += 64) { const int c = remap(ci[j]); lo = min(lo, c); hi = max(hi, c); }
        k += e - b;
    }
    lo = wave_min(lo); hi = wave_max(hi);
    if (lane == 0) { lo_out[w] = lo; hi_out[w] = hi; nnz_out[w] = k; }
}

// one thread per sampled row: entries, and entries on another 128-byte line of x than the previous entry of the row
__global__ void k_line_scatter(const int *rp, const int *ci, const int *rows, int n, int shift, RemapDev remap, unsigned long long *out)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int r = rows[i], b = rp[r], e = min(rp[r + 1], b + 512);
    if (e <= b) return;
    int prev = remap(ci[b]) >> shift, lines = 1;
    for (int j = b + 1; j < e; ++j) { const int l = remap(ci[j]) >> shift; lines += l != prev; prev = l; }
    atomicAdd(out, (unsigned long long)lines);
    atomicAdd(out + 1, (unsigned long long)(e - b));
}

// one thread per pair of equally long rows: positions whose columns lie within `within` of each other
__global__ void k_row_coherence(const int *rp, const int *ci, const int *rows, int npairs, RemapDev remap, unsigned long long *out, int within)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= npairs) return;
    const int a = rp[rows[2 * i]], b = rp[rows[2 * i + 1]], len = rp[rows[2 * i] + 1] - a;
    int near = 0;
    for (int k = 0; k < len; ++k) { const int d = remap(ci[a + k]) - remap(ci[b + k]); near += d > -within && d < within; }
    atomicAdd(out, (unsigned long long)near);
    atomicAdd(out + 1, (unsigned long long)len);
}

// one wave per medium block: first chunk (of the nchunks[b] the fill rule keeps) whose columns span more than 65534, and which of the
// chunks before it (< 64) span <= 254 columns (bit c of narrow[b]: one-byte ids, plan.cpp)
template <int K>
__global__ void k_chunk_spans(const int *rp, const int *ci, const int *ridM, const int *lenM, const int *nchunks, int nmed, int nb,
                              RemapDev remap, int *k16, unsigned long long *narrow)
{
    const int lane = threadIdx.x & 63, b = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (b >= nb) return;
    const int r0 = b * kMedRows, k = nchunks[b];
    int out = k;
    unsigned long long mask = 0;
    for (int c = 0; c < k; ++c) {
        int lo = 2147483647, hi = -1;
        // K columns x 16 rows = 16*K elements of the chunk, strided over the wave
        for (int e = lane; e < kMedRows * K; e += 64) {
            const int rr = e % kMedRows, kk = e / kMedRows, r = r0 + rr, i = c * K + kk;
            if (r < nmed && i < lenM[r]) { const int col = remap(ci[rp[ridM[r]] + i]); lo = min(lo, col); hi = max(hi, col); }
        }
        lo = wave_min(lo); hi = wave_max(hi);
        if (hi >= 0 && (long long)hi - lo > 65534) { out = c; break; }
        if (c < 64 && hi >= 0 && hi - lo <= 254) mask |= 1ull << c;
    }
    if (lane == 0) { k16[b] = out; if (narrow) narrow[b] = mask; }
}

template <class T>
__global__ void k_pack_long(const int *rp, const int *ci, const T *val, const int *ridL, const long long *startL, int nlong, RemapDev remap,
                            T *lv, int *lc)
{
    const int lane = threadIdx.x & 63, i = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (i >= nlong) return;
    const int r = ridL[i], a0 = rp[r], len = rp[r + 1] - a0;
    const long long at = startL[i], padded = startL[i + 1] - at;
    for (long long j = lane; j < padded; j += 64) {
        const bool in = j < len;
        lv[at + j] = in ? val[a0 + j] : (T)0;
        lc[at + j] = in ? remap(ci[a0 + j]) : -1;
    }
}

// 16-bit ids of the long pieces (plan.hpp long_cid16; the host packer's rule, plan.cpp): one wave per piece over its chunks of CH elements -- the chunk's smallest column is
// its base; a piece all of whose chunks span <= 65534 columns is narrow and keeps u16 offsets (0xFFFF = pad), a wide piece's u16 ids are 0
__global__ void k_long_cid16(const int *lc, const int *piece_ptr, int *piece_c16, int np, int CH, int *lbase, unsigned short *lc16)
{
    const int lane = threadIdx.x & 63, q = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (q >= np) return;
    const int e0 = piece_ptr[q], e1 = piece_ptr[q + 1], c0 = piece_c16[2 * q];
    bool narrow = true;
    for (int e = e0, c = c0; e < e1; e += CH, ++c) {
        int lo = 2147483647, hi = -1;
        for (int j = e + lane; j < min(e + CH, e1); j += 64) { const int col = lc[j]; if (col >= 0) { lo = min(lo, col); hi = max(hi, col); } }
        for (int d = 32; d > 0; d >>= 1) { lo = min(lo, __shfl_xor(lo, d)); hi = max(hi, __shfl_xor(hi, d)); }
        const int b = hi < 0 ? 0 : lo;
        if (lane == 0) lbase[c] = b;
        if (hi >= 0 && (long long)hi - lo > 65534) narrow = false;
        for (int j = e + lane; j < min(e + CH, e1); j += 64) { const int col = lc[j]; lc16[j] = col < 0 ? kLongPad16 : (unsigned short)(col - b); }      // (garbage where the chunk is wide: zeroed below)
    }
    narrow = narrow && e1 - e0 >= kLong16MinChunks * CH;
    if (lane == 0) piece_c16[2 * q + 1] = narrow ? 1 : 0;
    if (!narrow) for (int j = e0 + lane; j < e1; j += 64) lc16[j] = 0;
}

// one wave per medium block: regular chunks in lane-linear order (+ per-chunk base and u16 offsets in cid16 mode), then tails
template <class T, bool C16>
__global__ void k_pack_medium(const int *rp, const int *ci, const T *val, const int *ridM, const int *lenM, const int *med_ptr,
                              const int *irr_ptr, int nmed, int nb, RemapDev remap, T *mv, int *mc, unsigned short *mc16, int *mbase,
                              T *iv, int *ic, int pair_mode, const int *c8ptr, unsigned char *mc8, int *korig,
                              const int *win_cmin, const int *win_len, int bpw /* blocks per window; 0: per-chunk bases (Plan::win_rel16 off) */)
{
    constexpr int K = sizeof(T) == 8 ? 4 : 16, CH = kMedRows * K, VPL = CH / 64;
    const int lane = threadIdx.x & 63, b = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (b >= nb) return;
    const int c0 = med_ptr[b], nc = med_ptr[b + 1] - c0;
    const int r0 = b * kMedRows;
    const int nt_b = (irr_ptr[r0 + 1] - irr_ptr[r0] + K - 1) / K;
    const int npair = med_npair(nc, nt_b, (int)sizeof(T), pair_mode);      // r0 < nmed: a block has at least one row
    const bool oneshot = med_oneshot64(nc, nt_b);
    const int rr = lane & 15, kq = lane >> 4, r = r0 + rr;
    const bool row_ok = r < nmed;
    const int a0 = row_ok ? rp[ridM[r]] : 0, len = row_ok ? lenM[r] : 0;
    // cid16 mode: n8 narrow chunks of the paired region go to its front (one-byte ids), the others follow in their order (plan.cpp packs the same)
    int n8 = 0, a8 = 0, a16 = 0; size_t e8 = 0, e16 = (size_t)c0 * CH;
    if constexpr (C16) { n8 = c8ptr[b + 1] - c8ptr[b]; e8 = (size_t)c8ptr[b] * CH; e16 = ((size_t)c0 - (size_t)c8ptr[b]) * CH; }
    for (int c = 0; c < nc; ++c) {
        int col[VPL]; T v[VPL];
        int lo = 2147483647, hi = -1;
#pragma unroll
        for (int q = 0; q < VPL; ++q) {
            // f64: lane = kk*16 + rr holds entry kk of the chunk ; f16: lane = kq*16 + rr holds entries 4kq .. 4kq+3
            const int i = c * K + (VPL == 1 ? kq : 4 * kq + q);
            const bool in = row_ok && i < len;
            v[q] = in ? val[a0 + i] : (T)0;
            col[q] = in ? remap(ci[a0 + i]) : -1;
            if (in) { lo = min(lo, col[q]); hi = max(hi, col[q]); }
        }
        if constexpr (C16) {
            lo = wave_min(lo); hi = wave_max(hi);
            const bool narrow = c < npair && c < 64 && a8 < n8 && hi >= 0 && hi - lo <= 254;       // wave-uniform
            if (lo == 2147483647) lo = 0;
            if (bpw > 0 && win_len[b / bpw] > 0) lo = win_cmin[b / bpw];          // window-relative offsets (plan.cpp packs the same)
            const int pos = c >= npair ? c : narrow ? a8++ : n8 + a16++;
            if (lane == 0) { mbase[c0 + pos] = lo; korig[c0 + pos] = c; }
            const size_t at = (size_t)c0 * CH + med_elem_index(npair, pos, lane, 0, VPL, CH);     // pipelined blocks: pairs of chunks interleaved per lane (plan.hpp)
#pragma unroll
            for (int q = 0; q < VPL; ++q) {
                mv[at + q] = v[q];
                if (pos < n8) mc8[e8 + med_cid8_index(pos, lane, CH, oneshot)] = col[q] < 0 ? (unsigned char)0xFF : (unsigned char)(col[q] - lo);      // f64 only: VPL = 1
                else mc16[e16 + med_elem_index(npair - n8, pos - n8, lane, q, VPL, CH)] = col[q] < 0 ? (unsigned short)0xFFFF : (unsigned short)(col[q] - lo);
            }
        } else {
            const size_t at = (size_t)c0 * CH + med_elem_index(npair, c, lane, 0, VPL, CH);
#pragma unroll
            for (int q = 0; q < VPL; ++q) { mv[at + q] = v[q]; mc[at + q] = col[q]; }
        }
    }
    // irregular tail = the LAST tl entries of the row (dasp_f64.h:1094-1106); lanes kq = 0..3 of a row share the copy
    if (row_ok) {
        const int t0 = irr_ptr[r], tl = irr_ptr[r + 1] - t0;
        for (int j = kq; j < tl; j += 4) { iv[t0 + j] = val[a0 + len - tl + j]; ic[t0 + j] = remap(ci[a0 + len - tl + j]); }
    }
}

// short slabs: one thread per (slab position, k); positions past the slab's row count are pads
template <class T>
__global__ void k_pack_short(const int *rp, const int *ci, const T *val, const int *list, int count, int tiles, int L, long long elem_off,
                             int SR, int seg, RemapDev remap, T *sv, int *sc)
{
    const long long n = (long long)tiles * short_tile_elems(seg != 0, L, SR);
    for (long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x; e < n; e += (long long)gridDim.x * blockDim.x) {
        long long t; int k; bool in;
        if (seg) {      // the inverse of plan.hpp short_elem_index: element e = tile * 64 + lane
            const int lane = (int)(e & 63), sub = lane & 15, per16 = 16 / L, rloc = sub / L;
            k = sub % L;
            t = (e >> 6) * (4 * per16) + (lane >> 4) * per16 + rloc;
            in = rloc < per16 && t < count;
        } else {
            const long long tile = e / ((long long)L * SR);
            k = (int)((e / SR) % L);
            t = tile * SR + (int)(e % SR);
            in = t < count;
        }
        const int a0 = in ? rp[list[t]] : 0;
        sv[elem_off + e] = in ? val[a0 + k] : (T)0;
        sc[elem_off + e] = in ? remap(ci[a0 + k]) : -1;
    }
}

// small helper: a device copy of a host vector, freed on scope exit
template <class U>
struct DevVec {
    U *d = nullptr;
    template <class A>
    int init(const std::vector<U, A> &h)
    {
        if (h.empty()) return DASP_OK;
        if (hipMalloc(&d, h.size() * sizeof(U)) != hipSuccess) { set_error("hipMalloc (device packing scratch)"); return DASP_ERR_HIP; }
        if (hipMemcpy(d, h.data(), h.size() * sizeof(U), hipMemcpyHostToDevice) != hipSuccess) { set_error("hipMemcpy (device packing scratch)"); return DASP_ERR_HIP; }
        return DASP_OK;
    }
    ~DevVec() { if (d) (void)hipFree(d); }
};

// uninitialised device scratch of n elements, freed on scope exit
template <class U>
struct DevBuf {
    U *d = nullptr;
    int init(size_t n)
    {
        if (hipMalloc(&d, std::max<size_t>(n, 1) * sizeof(U)) != hipSuccess) { d = nullptr; set_error("hipMalloc (device packing scratch)"); return DASP_ERR_HIP; }
        return DASP_OK;
    }
    ~DevBuf() { if (d) (void)hipFree(d); }
};

struct RemapHolder {
    DevVec<int> bounds;
    RemapDev r{nullptr, 0, 0};
    int init(const Plan &p)
    {
        if (p.opt.n_parts <= 0) return DASP_OK;
        if (int rc = bounds.init(p.part_bounds)) return rc;
        r = RemapDev{bounds.d, p.opt.n_parts, p.opt.part_stride};
        return DASP_OK;
    }
};

inline int waves_grid(int units) { return (units + 3) / 4; }   // 4 waves per 256-thread workgroup

}  // namespace

int devpack_validate(const Plan &p, const DevCsr &d)
{
    if (p.nnz == 0) return DASP_OK;
    DevBuf<int> bad; if (int rc = bad.init(1)) return rc;
    int h = 0;
    HIP_TRYP(hipMemset(bad.d, 0, sizeof(int)));
    hipLaunchKernelGGL(k_validate, dim3(2048), dim3(256), 0, 0, d.ci, (long long)p.nnz, p.n, bad.d);
    HIP_TRYP(hipGetLastError());            // a failed launch would leave `bad` at 0 and wave every id through
    HIP_TRYP(hipMemcpy(&h, bad.d, sizeof(int), hipMemcpyDeviceToHost));
    if (h) { set_error("column index out of range"); return DASP_ERR_ARG; }
    return DASP_OK;
}

int devpack_window_spans(const Plan &p, const DevCsr &d, const raw_vector<int> &ridW, int R, int *lo, int *hi, long long *wnnz)
{
    const int nmed = (int)ridW.size(), nW = (nmed + R - 1) / R;
    if (nW == 0) return DASP_OK;
    RemapHolder rm; if (int rc = rm.init(p)) return rc;
    DevVec<int> dr; if (int rc = dr.init(ridW)) return rc;
    DevBuf<int> dlo; DevBuf<long long> dn;
    if (int rc = dlo.init(2 * (size_t)nW)) return rc;
    if (int rc = dn.init((size_t)nW)) return rc;
    hipLaunchKernelGGL(k_window_spans, dim3(waves_grid(nW)), dim3(256), 0, 0, d.rp, d.ci, dr.d, nmed, R, rm.r, dlo.d, dlo.d + nW, dn.d, nW);
    HIP_TRYP(hipGetLastError());
    HIP_TRYP(hipMemcpy(lo, dlo.d, sizeof(int) * (size_t)nW, hipMemcpyDeviceToHost));
    HIP_TRYP(hipMemcpy(hi, dlo.d + nW, sizeof(int) * (size_t)nW, hipMemcpyDeviceToHost));
    HIP_TRYP(hipMemcpy(wnnz, dn.d, sizeof(long long) * (size_t)nW, hipMemcpyDeviceToHost));
    return DASP_OK;
}

int devpack_line_scatter(const Plan &p, const DevCsr &d, const std::vector<int> &rows, long long *lines, long long *entries)
{
    *lines = 0; *entries = 0;
    if (rows.empty()) return DASP_OK;
    RemapHolder rm; if (int rc = rm.init(p)) return rc;
    DevVec<int> dr; if (int rc = dr.init(rows)) return rc;
    unsigned long long h[2] = {0, 0};
    DevBuf<unsigned long long> dout; if (int rc = dout.init(2)) return rc;
    HIP_TRYP(hipMemset(dout.d, 0, sizeof h));
    const int n = (int)rows.size();
    hipLaunchKernelGGL(k_line_scatter, dim3((n + 255) / 256), dim3(256), 0, 0, d.rp, d.ci, dr.d, n, p.geo.vbytes == 8 ? 4 : 6, rm.r, dout.d);
    HIP_TRYP(hipGetLastError());
    HIP_TRYP(hipMemcpy(h, dout.d, sizeof h, hipMemcpyDeviceToHost));
    *lines = (long long)h[0]; *entries = (long long)h[1];
    return DASP_OK;
}

int devpack_row_coherence(const Plan &p, const DevCsr &d, const std::vector<int> &rows, int within, long long *near, long long *entries)
{
    *near = 0; *entries = 0;
    const int npairs = (int)(rows.size() / 2);
    if (npairs == 0) return DASP_OK;
    RemapHolder rm; if (int rc = rm.init(p)) return rc;
    DevVec<int> dr; if (int rc = dr.init(rows)) return rc;
    unsigned long long h[2] = {0, 0};
    DevBuf<unsigned long long> dout; if (int rc = dout.init(2)) return rc;
    HIP_TRYP(hipMemset(dout.d, 0, sizeof h));
    hipLaunchKernelGGL(k_row_coherence, dim3((npairs + 255) / 256), dim3(256), 0, 0, d.rp, d.ci, dr.d, npairs, rm.r, dout.d, within);
    HIP_TRYP(hipGetLastError());
    HIP_TRYP(hipMemcpy(h, dout.d, sizeof h, hipMemcpyDeviceToHost));
    *near = (long long)h[0]; *entries = (long long)h[1];
    return DASP_OK;
}

int devpack_chunk_spans(const Plan &p, const DevCsr &d, const raw_vector<int> &ridM, const raw_vector<int> &lenM,
                        const std::vector<int> &nchunks, int *k16, unsigned long long *narrow_mask)
{
    const int nmed = (int)ridM.size(), nb = (nmed + kMedRows - 1) / kMedRows;
    if (nb == 0) return DASP_OK;
    RemapHolder rm; if (int rc = rm.init(p)) return rc;
    DevVec<int> dr, dl, dk;
    if (int rc = dr.init(ridM)) return rc;
    if (int rc = dl.init(lenM)) return rc;
    if (int rc = dk.init(nchunks)) return rc;
    DevBuf<int> dout; if (int rc = dout.init((size_t)nb)) return rc;
    DevBuf<unsigned long long> dmask; if (narrow_mask) if (int rc = dmask.init((size_t)nb)) return rc;
    if (p.precision == 64)
        hipLaunchKernelGGL((k_chunk_spans<4>), dim3(waves_grid(nb)), dim3(256), 0, 0, d.rp, d.ci, dr.d, dl.d, dk.d, nmed, nb, rm.r, dout.d, narrow_mask ? dmask.d : nullptr);
    else
        hipLaunchKernelGGL((k_chunk_spans<16>), dim3(waves_grid(nb)), dim3(256), 0, 0, d.rp, d.ci, dr.d, dl.d, dk.d, nmed, nb, rm.r, dout.d, narrow_mask ? dmask.d : nullptr);
    HIP_TRYP(hipGetLastError());
    HIP_TRYP(hipMemcpy(k16, dout.d, sizeof(int) * (size_t)nb, hipMemcpyDeviceToHost));
    if (narrow_mask) HIP_TRYP(hipMemcpy(narrow_mask, dmask.d, sizeof(unsigned long long) * (size_t)nb, hipMemcpyDeviceToHost));
    return DASP_OK;
}

template <class T>
static int pack_all_typed(Plan &p, const DevCsr &d, const PackMeta &m)
{
    DevicePlan &dp = *p.dev;
    char *base = static_cast<char *>(dp.arena);
    const T *val = static_cast<const T *>(d.val);
    RemapHolder rm; if (int rc = rm.init(p)) return rc;
    const int nlong = (int)m.ridL->size(), nmed = (int)m.ridM->size(), nb = (nmed + kMedRows - 1) / kMedRows;
    if (nlong > 0) {
        DevVec<int> dr; DevVec<long long> ds;
        if (int rc = dr.init(*m.ridL)) return rc;
        if (int rc = ds.init(*m.startL)) return rc;
        hipLaunchKernelGGL((k_pack_long<T>), dim3(waves_grid(nlong)), dim3(256), 0, 0, d.rp, d.ci, val, dr.d, ds.d, nlong, rm.r,
                           reinterpret_cast<T *>(base + dp.map.long_val), reinterpret_cast<int *>(base + dp.map.long_cid));
        HIP_TRYP(hipGetLastError());
        const int np = (int)p.piece_dst.size();
        if (np > 0) {
            int *pc16 = const_cast<int *>(dp.args.piece_c16);
            hipLaunchKernelGGL(k_long_cid16, dim3(waves_grid(np)), dim3(256), 0, 0, reinterpret_cast<const int *>(base + dp.map.long_cid), dp.args.piece_ptr, pc16, np, p.geo.chunk,
                               reinterpret_cast<int *>(base + dp.map.long_base), reinterpret_cast<unsigned short *>(base + dp.map.long_cid16));
            HIP_TRYP(hipGetLastError());
            HIP_TRYP(hipMemcpy(p.piece_c16.data(), pc16, p.piece_c16.size() * sizeof(int), hipMemcpyDeviceToHost));      // which pieces are narrow: back to the host (plan files, decoders)
            choose_long16(p);
        }
        HIP_TRYP(hipDeviceSynchronize());
    }
    if (nb > 0) {
        DevVec<int> dr, dl;
        if (int rc = dr.init(*m.ridM)) return rc;
        if (int rc = dl.init(*m.lenM)) return rc;
        T *mv = reinterpret_cast<T *>(base + dp.map.med_val), *iv = reinterpret_cast<T *>(base + dp.map.irr_val);
        int *mc = reinterpret_cast<int *>(base + dp.map.med_cid), *ic = reinterpret_cast<int *>(base + dp.map.irr_cid);
        unsigned short *mc16 = reinterpret_cast<unsigned short *>(base + dp.map.med_cid16);
        int *mb = reinterpret_cast<int *>(base + dp.map.med_base);
        unsigned char *mc8 = reinterpret_cast<unsigned char *>(base + dp.map.med_cid8);
        DevBuf<int> dko;                                    // which chunk of its block sits at each position: back to the host (plan files, decoders)
        if (p.cid16) if (int rc = dko.init(std::max<size_t>(1, p.med_korig.size()))) return rc;
        if (p.cid16)
            hipLaunchKernelGGL((k_pack_medium<T, true>), dim3(waves_grid(nb)), dim3(256), 0, 0, d.rp, d.ci, val, dr.d, dl.d, dp.args.med_ptr,
                               dp.args.irr_ptr, nmed, nb, rm.r, mv, mc, mc16, mb, iv, ic, p.pair_mode, dp.args.med_c8ptr, mc8, dko.d,
                               dp.args.win_cmin, dp.args.win_len, p.win_rel16 ? p.row_window / kMedRows : 0);
        else
            hipLaunchKernelGGL((k_pack_medium<T, false>), dim3(waves_grid(nb)), dim3(256), 0, 0, d.rp, d.ci, val, dr.d, dl.d, dp.args.med_ptr,
                               dp.args.irr_ptr, nmed, nb, rm.r, mv, mc, mc16, mb, iv, ic, p.pair_mode, nullptr, nullptr, nullptr, nullptr, nullptr, 0);
        HIP_TRYP(hipGetLastError());
        HIP_TRYP(hipDeviceSynchronize());
        if (p.cid16 && !p.med_korig.empty()) HIP_TRYP(hipMemcpy(p.med_korig.data(), dko.d, p.med_korig.size() * sizeof(int), hipMemcpyDeviceToHost));
    }
    for (int g = 0; g < kNumShortGroups; ++g) {
        const ShortGroup &G = p.grp[g];
        if (G.tiles == 0 || G.len == 0) continue;
        DevVec<int> dl;
        if (int rc = dl.init(*m.glist[g])) return rc;
        const long long n = (long long)G.tiles * short_tile_elems(G.seg != 0, G.len, p.geo.short_rows);
        hipLaunchKernelGGL((k_pack_short<T>), dim3((unsigned)std::min<long long>((n + 255) / 256, 65535)), dim3(256), 0, 0, d.rp, d.ci, val,
                           dl.d, G.count, G.tiles, G.len, G.elem_off, p.geo.short_rows, G.seg, rm.r,
                           reinterpret_cast<T *>(base + dp.map.short_val), reinterpret_cast<int *>(base + dp.map.short_cid));
        HIP_TRYP(hipGetLastError());
        HIP_TRYP(hipDeviceSynchronize());
    }
    HIP_TRYP(hipGetLastError());
    return DASP_OK;
}

// ---- column panels of a device-resident CSR (plan.cpp build_panels): the split by column range as two kernels, one wave per row.
// Lane k (k < P <= 64) holds panel k's counter / write cursor; a chunk of 64 entries is ranked per panel with ballots, so the entries of
// a row keep their order inside every panel -- the sub-matrices are exactly those of the host split.
__device__ __forceinline__ int panel_of_dev(const int *bnd, int P, int c)
{
    int lo = 0, hi = P;                        // last k with bnd[k] <= c
    while (hi - lo > 1) { const int mid = (lo + hi) >> 1; if (bnd[mid] <= c) lo = mid; else hi = mid; }
    return lo;
}
__global__ void k_panel_count(const int *rp, const int *ci, int m, RemapDev remap, const int *bnd, int P, int *cnt /* [P][m + 1] */)
{
    const int lane = threadIdx.x & 63, i = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (i >= m) return;
    const int a0 = rp[i], a1 = rp[i + 1];
    int mine = 0;
    for (int j0 = a0; j0 < a1; j0 += 64) {
        const int j = j0 + lane;
        const int k = j < a1 ? panel_of_dev(bnd, P, remap(ci[j])) : -1;
        for (int q = 0; q < P; ++q) {
            const int tot = __popcll(__ballot(k == q));
            if (lane == q) mine += tot;
        }
    }
    if (lane < P) cnt[(size_t)lane * ((size_t)m + 1) + (size_t)i + 1] = mine;
}
template <class T>
__global__ void k_panel_scatter(const int *rp, const int *ci, const T *val, int m, RemapDev remap, const int *bnd, int P,
                                const int *rpP /* [P][m + 1], scanned */, int *const *ciP, T *const *valP)
{
    const int lane = threadIdx.x & 63, i = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (i >= m) return;
    const int a0 = rp[i], a1 = rp[i + 1];
    int cursor = lane < P ? rpP[(size_t)lane * ((size_t)m + 1) + (size_t)i] : 0;
    const unsigned long long below = lane == 0 ? 0ull : (~0ull >> (64 - lane));
    for (int j0 = a0; j0 < a1; j0 += 64) {
        const int j = j0 + lane;
        const bool in = j < a1;
        const int c = in ? remap(ci[j]) : 0;
        const int k = in ? panel_of_dev(bnd, P, c) : -1;
        int at = 0;
        for (int q = 0; q < P; ++q) {
            const unsigned long long mask = __ballot(k == q);
            const int base = __shfl(cursor, q);
            if (k == q) at = base + __popcll(mask & below);
            if (lane == q) cursor += __popcll(mask);
        }
        if (in) { ciP[k][at] = c; valP[k][at] = val[j]; }
    }
}

// splits the device CSR `d` of plan `p` into P column ranges [bnd[k], bnd[k + 1]).  rpP_host[k] = the panel's row pointer (host copy);
// out[k] = its device CSR (columns already remapped).  The device arrays are owned by `keep` (freed with it).
int devpack_panel_split(const Plan &p, const DevCsr &d, const std::vector<int> &bnd, int P, std::vector<std::vector<int>> &rpP_host,
                        std::vector<DevCsr> &out, std::vector<std::shared_ptr<void>> &keep)
{
    const int m = p.m;
    const size_t vb = (size_t)p.geo.vbytes, row = (size_t)m + 1;
    auto dmalloc = [&](size_t bytes, void **ptr) -> int {
        if (hipMalloc(ptr, std::max<size_t>(bytes, 16)) != hipSuccess) { set_error("hipMalloc (column-panel split)"); return DASP_ERR_HIP; }
        keep.emplace_back(*ptr, [](void *q) { (void)hipFree(q); });
        return DASP_OK;
    };
    RemapHolder rm; if (int rc = rm.init(p)) return rc;
    DevVec<int> dbnd; if (int rc = dbnd.init(bnd)) return rc;
    void *cnt = nullptr;
    if (int rc = dmalloc(row * (size_t)P * sizeof(int), &cnt)) return rc;
    HIP_TRYP(hipMemset(cnt, 0, row * (size_t)P * sizeof(int)));
    if (m > 0) hipLaunchKernelGGL(k_panel_count, dim3(waves_grid(m)), dim3(256), 0, 0, d.rp, d.ci, m, rm.r, dbnd.d, P, static_cast<int *>(cnt));
    HIP_TRYP(hipGetLastError());
    // the panels' row pointers: an inclusive scan of each panel's counts in place on the device (element 0 of a panel is 0), then every
    // panel's pointer comes to the host in a thread of its own -- the host stages of the panel builds read it.  (First version: all counts
    // to the host, serial scans there, all pointers back: 37 of ljournal-2008's ~120 ms, most of it pageable copies of 4 x 21 MB each way.)
    {
        size_t tmp_bytes = 0;
        int *c0 = static_cast<int *>(cnt);
        if (hipcub::DeviceScan::InclusiveSum(nullptr, tmp_bytes, c0, c0, (int)row) != hipSuccess) { set_error("hipcub scan (column-panel split)"); return DASP_ERR_HIP; }
        void *tmp = nullptr;
        if (int rc = dmalloc(tmp_bytes, &tmp)) return rc;
        for (int k = 0; k < P; ++k) {
            int *ck = c0 + (size_t)k * row;
            if (hipcub::DeviceScan::InclusiveSum(tmp, tmp_bytes, ck, ck, (int)row) != hipSuccess) { set_error("hipcub scan (column-panel split)"); return DASP_ERR_HIP; }
        }
        HIP_TRYP(hipDeviceSynchronize());
    }
    rpP_host.assign((size_t)P, std::vector<int>());
    {
        int device = 0;
        HIP_TRYP(hipGetDevice(&device));
        std::vector<int> bad((size_t)P, 0);
        std::vector<std::thread> th;
        for (int k = 0; k < P; ++k)
            th.emplace_back([&, k] {
                (void)hipSetDevice(device);
                rpP_host[(size_t)k].resize(row);
                bad[(size_t)k] = hipMemcpy(rpP_host[(size_t)k].data(), static_cast<const int *>(cnt) + (size_t)k * row, row * sizeof(int), hipMemcpyDeviceToHost) != hipSuccess;
            });
        for (auto &t : th) t.join();
        for (int k = 0; k < P; ++k) if (bad[(size_t)k]) { (void)hipGetLastError(); set_error("hipMemcpy (column-panel row pointers)"); return DASP_ERR_HIP; }
    }
    std::vector<int *> h_ci((size_t)P, nullptr);
    std::vector<void *> h_val((size_t)P, nullptr);
    out.assign((size_t)P, DevCsr{nullptr, nullptr, nullptr});
    for (int k = 0; k < P; ++k) {
        const size_t nk = (size_t)rpP_host[(size_t)k][(size_t)m];
        void *a = nullptr, *b = nullptr;
        if (int rc = dmalloc(nk * sizeof(int), &a)) return rc;
        if (int rc = dmalloc(nk * vb, &b)) return rc;
        h_ci[(size_t)k] = static_cast<int *>(a); h_val[(size_t)k] = b;
        out[(size_t)k] = DevCsr{static_cast<const int *>(cnt) + (size_t)k * row, static_cast<const int *>(a), b};
    }
    void *pc = nullptr, *pv = nullptr;
    if (int rc = dmalloc((size_t)P * sizeof(void *), &pc)) return rc;
    if (int rc = dmalloc((size_t)P * sizeof(void *), &pv)) return rc;
    HIP_TRYP(hipMemcpy(pc, h_ci.data(), (size_t)P * sizeof(void *), hipMemcpyHostToDevice));
    HIP_TRYP(hipMemcpy(pv, h_val.data(), (size_t)P * sizeof(void *), hipMemcpyHostToDevice));
    if (m > 0) {
        if (p.precision == 64)
            hipLaunchKernelGGL((k_panel_scatter<double>), dim3(waves_grid(m)), dim3(256), 0, 0, d.rp, d.ci, static_cast<const double *>(d.val), m, rm.r, dbnd.d, P,
                               static_cast<const int *>(cnt), static_cast<int *const *>(pc), static_cast<double *const *>(pv));
        else
            hipLaunchKernelGGL((k_panel_scatter<_Float16>), dim3(waves_grid(m)), dim3(256), 0, 0, d.rp, d.ci, static_cast<const _Float16 *>(d.val), m, rm.r, dbnd.d, P,
                               static_cast<const int *>(cnt), static_cast<int *const *>(pc), static_cast<_Float16 *const *>(pv));
    }
    HIP_TRYP(hipGetLastError());
    HIP_TRYP(hipDeviceSynchronize());
    return DASP_OK;
}

// ---- opt.sort_columns on a device CSR
__global__ void k_rows_unsorted(const int *rp, const int *ci, int m, int *flag)
{
    const int lane = threadIdx.x & 63, i = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (i >= m) return;
    const int a0 = rp[i], a1 = rp[i + 1];
    bool bad = false;
    for (int j = a0 + 1 + lane; j < a1; j += 64) bad = bad || ci[j] < ci[j - 1];
    if (__any(bad) && lane == 0) *flag = 1;
}
int devpack_sort_columns(const Plan &p, const DevCsr &d, std::vector<std::shared_ptr<void>> &keep, DevCsr *out, int *did)
{
    *did = 0;
    const int m = p.m, nnz = p.nnz;
    if (m <= 0 || nnz <= 0) return DASP_OK;
    auto dmalloc = [&](size_t bytes, void **ptr) -> int {
        if (hipMalloc(ptr, std::max<size_t>(bytes, 16)) != hipSuccess) { set_error("hipMalloc (sort_columns)"); return DASP_ERR_HIP; }
        keep.emplace_back(*ptr, [](void *q) { (void)hipFree(q); });
        return DASP_OK;
    };
    void *flag = nullptr;
    if (int rc = dmalloc(16, &flag)) return rc;
    HIP_TRYP(hipMemset(flag, 0, 16));
    hipLaunchKernelGGL(k_rows_unsorted, dim3(waves_grid(m)), dim3(256), 0, 0, d.rp, d.ci, m, static_cast<int *>(flag));
    HIP_TRYP(hipGetLastError());
    int h = 0;
    HIP_TRYP(hipMemcpy(&h, flag, sizeof(int), hipMemcpyDeviceToHost));
    if (!h) return DASP_OK;
    const size_t vb = (size_t)p.geo.vbytes;
    void *ci2 = nullptr, *val2 = nullptr, *tmp = nullptr;
    if (int rc = dmalloc((size_t)nnz * 4, &ci2)) return rc;
    if (int rc = dmalloc((size_t)nnz * vb, &val2)) return rc;
    size_t tmp_bytes = 0;
    hipError_t e;
    // the values ride along as plain bit patterns
    if (vb == 8) e = hipcub::DeviceSegmentedSort::StableSortPairs(nullptr, tmp_bytes, d.ci, static_cast<int *>(ci2), static_cast<const unsigned long long *>(d.val), static_cast<unsigned long long *>(val2), nnz, m, d.rp, d.rp + 1);
    else e = hipcub::DeviceSegmentedSort::StableSortPairs(nullptr, tmp_bytes, d.ci, static_cast<int *>(ci2), static_cast<const unsigned short *>(d.val), static_cast<unsigned short *>(val2), nnz, m, d.rp, d.rp + 1);
    if (e != hipSuccess) { set_error("hipcub segmented sort (sort_columns)"); return DASP_ERR_HIP; }
    if (int rc = dmalloc(tmp_bytes, &tmp)) return rc;
    if (vb == 8) e = hipcub::DeviceSegmentedSort::StableSortPairs(tmp, tmp_bytes, d.ci, static_cast<int *>(ci2), static_cast<const unsigned long long *>(d.val), static_cast<unsigned long long *>(val2), nnz, m, d.rp, d.rp + 1);
    else e = hipcub::DeviceSegmentedSort::StableSortPairs(tmp, tmp_bytes, d.ci, static_cast<int *>(ci2), static_cast<const unsigned short *>(d.val), static_cast<unsigned short *>(val2), nnz, m, d.rp, d.rp + 1);
    if (e != hipSuccess) { set_error("hipcub segmented sort (sort_columns)"); return DASP_ERR_HIP; }
    HIP_TRYP(hipDeviceSynchronize());
    *out = DevCsr{d.rp, static_cast<const int *>(ci2), val2};
    *did = 1;
    return DASP_OK;
}

// ---- row tiles of a column panel (plan.cpp build_panels): one wave per row moves the row either into the tiles' arrays or into the
// sub-matrix of the rows that stay with the panel's plan; a row's elements keep their order
template <class T>
__global__ void k_row_tiles_move(const int *rp, const int *ci, const T *val, int m, const int *rp_rest, const int *at, int *rest_ci, T *rest_val, int *rt_cid, T *rt_val)
{
    const int lane = threadIdx.x & 63, i = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (i >= m) return;
    const int a0 = rp[i], n = rp[i + 1] - a0, t = at[i];
    int *dc = t >= 0 ? rt_cid + t : rest_ci + rp_rest[i];
    T *dv = t >= 0 ? rt_val + t : rest_val + rp_rest[i];
    for (int j = lane; j < n; j += 64) { dc[j] = ci[a0 + j]; dv[j] = val[a0 + j]; }
}
int devpack_row_tiles(const Plan &p, DevCsr &panel, const std::vector<int> &rp_rest, const std::vector<int> &at, size_t cnt,
                      std::vector<std::shared_ptr<void>> &keep, DevRowTiles *out)
{
    const int m = p.m;
    const size_t vb = (size_t)p.geo.vbytes, n_rest = (size_t)rp_rest[(size_t)m];
    auto dmalloc = [&](size_t bytes, void **ptr) -> int {
        if (hipMalloc(ptr, std::max<size_t>(bytes, 16)) != hipSuccess) { set_error("hipMalloc (row tiles)"); return DASP_ERR_HIP; }
        keep.emplace_back(*ptr, [](void *q) { (void)hipFree(q); });
        return DASP_OK;
    };
    void *d_rp = nullptr, *d_at = nullptr, *r_ci = nullptr, *r_val = nullptr, *t_ci = nullptr, *t_val = nullptr;
    if (int rc = dmalloc(((size_t)m + 1) * 4, &d_rp)) return rc;
    if (int rc = dmalloc((size_t)m * 4, &d_at)) return rc;
    if (int rc = dmalloc(n_rest * 4, &r_ci)) return rc;
    if (int rc = dmalloc(n_rest * vb, &r_val)) return rc;
    if (int rc = dmalloc(cnt * 4, &t_ci)) return rc;
    if (int rc = dmalloc(cnt * vb, &t_val)) return rc;
    HIP_TRYP(hipMemcpy(d_rp, rp_rest.data(), ((size_t)m + 1) * 4, hipMemcpyHostToDevice));
    if (m > 0) {
        HIP_TRYP(hipMemcpy(d_at, at.data(), (size_t)m * 4, hipMemcpyHostToDevice));
        if (p.precision == 64)
            hipLaunchKernelGGL((k_row_tiles_move<double>), dim3(waves_grid(m)), dim3(256), 0, 0, panel.rp, panel.ci, static_cast<const double *>(panel.val), m, static_cast<const int *>(d_rp),
                               static_cast<const int *>(d_at), static_cast<int *>(r_ci), static_cast<double *>(r_val), static_cast<int *>(t_ci), static_cast<double *>(t_val));
        else
            hipLaunchKernelGGL((k_row_tiles_move<_Float16>), dim3(waves_grid(m)), dim3(256), 0, 0, panel.rp, panel.ci, static_cast<const _Float16 *>(panel.val), m, static_cast<const int *>(d_rp),
                               static_cast<const int *>(d_at), static_cast<int *>(r_ci), static_cast<_Float16 *>(r_val), static_cast<int *>(t_ci), static_cast<_Float16 *>(t_val));
        HIP_TRYP(hipGetLastError());
        HIP_TRYP(hipDeviceSynchronize());
    }
    panel = DevCsr{static_cast<const int *>(d_rp), static_cast<const int *>(r_ci), r_val};
    out->val = t_val; out->cid = static_cast<int *>(t_ci);
    return DASP_OK;
}
int devpack_place_row_tiles(Plan &q, const DevRowTiles &src)
{
    if (!q.dev || !q.dev->arena) { set_error("panel plan not on the device"); return DASP_ERR_STATE; }
    char *base = static_cast<char *>(q.dev->arena);
    HIP_TRYP(hipMemcpy(base + q.dev->map.rt_val, src.val, q.cnt_rt * (size_t)q.geo.vbytes, hipMemcpyDeviceToDevice));
    HIP_TRYP(hipMemcpy(base + q.dev->map.rt_cid, src.cid, q.cnt_rt * 4, hipMemcpyDeviceToDevice));
    return DASP_OK;
}

// remapped column ids at arbitrary nonzero positions, back on the host: the samples the automatic column-panel rule looks at
__global__ void k_gather_cols(const int *ci, const long long *idx, long long n, long long start, long long stride, RemapDev remap, int *out)
{
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x)
        out[i] = remap(ci[idx ? idx[i] : start + i * stride]);
}
int devpack_gather_columns(const Plan &p, const DevCsr &d, const std::vector<long long> *idx, long long start, long long stride, long long count, std::vector<int> &out)
{
    const long long n = idx ? (long long)idx->size() : count;
    out.assign((size_t)std::max<long long>(n, 0), 0);
    if (n <= 0) return DASP_OK;
    RemapHolder rm; if (int rc = rm.init(p)) return rc;
    DevVec<long long> di;
    if (idx) { if (int rc = di.init(*idx)) return rc; }
    DevBuf<int> dout; if (int rc = dout.init((size_t)n)) return rc;
    hipLaunchKernelGGL(k_gather_cols, dim3((unsigned)std::min<long long>(4096, (n + 255) / 256)), dim3(256), 0, 0, d.ci, idx ? di.d : nullptr, n, start, stride, rm.r, dout.d);
    HIP_TRYP(hipGetLastError());
    HIP_TRYP(hipMemcpy(out.data(), dout.d, (size_t)n * sizeof(int), hipMemcpyDeviceToHost));
    return DASP_OK;
}

int devpack_finish_panels(Plan &p) { return upload_plan(p); }
// the two-phase form of a device-resident CSR: its tile sort runs on the host for now (preprocessing, not the hot path) -- the column ids and values
// are copied over once, the plan is packed by build_two_phase and uploaded like any host-built plan
int devpack_fetch_csr(const Plan &p, const DevCsr &d, int *ci, void *val)
{
    if (p.nnz <= 0) return DASP_OK;
    HIP_TRYP(hipMemcpy(ci, d.ci, (size_t)p.nnz * sizeof(int), hipMemcpyDeviceToHost));
    HIP_TRYP(hipMemcpy(val, d.val, (size_t)p.nnz * (size_t)p.geo.vbytes, hipMemcpyDeviceToHost));
    return DASP_OK;
}
int devpack_current_device() { int d = 0; if (hipGetDevice(&d) != hipSuccess) { (void)hipGetLastError(); d = 0; } return d; }
void devpack_use_device(int device) { if (device >= 0) (void)hipSetDevice(device); }

int devpack_all(Plan &p, const DevCsr &d, const PackMeta &m)
{
    // arena + every O(rows) array through the normal upload path (the nnz-sized regions are left unwritten) ...
    if (int rc = upload_plan_unpacked(p)) return rc;
    // ... then the kernels above fill those regions straight from the device CSR
    const int rc = p.precision == 64 ? pack_all_typed<double>(p, d, m) : pack_all_typed<_Float16>(p, d, m);
    if (rc == DASP_OK) p.host_dropped = true;       // no host copies of the packed nonzeros exist
    return rc;
}

// test hook of the multi-GPU choreography (multigpu.cpp, DASP_MG_FAKE_ALLGATHER_US): a kernel that occupies `stream` for ~micros
// microseconds (one lane polling the 100 MHz wall clock), standing in for the duration of an all-gather on a one-GPU box
__global__ void k_spin_us(long long ticks)
{
    const long long t0 = wall_clock64();
    while (wall_clock64() - t0 < ticks) __builtin_amdgcn_s_sleep(8);
}
// the same with the footprint of RCCL's collective kernels on gfx950 (rcclGenericKernel<1|2|4> in this image's librccl.so: 256 threads,
// .vgpr_count 261-280, 19 744 B of LDS, one workgroup per channel): what has to find room beside the product kernels for the exchange to
// overlap them at all.  Every workgroup spins from ITS OWN start, so a late start ends the "exchange" late, as it would RCCL's.
__global__ void __launch_bounds__(256) k_spin_fat(long long ticks, int *sink)
{
    __shared__ int pad[19744 / 4];
    asm volatile("v_mov_b32 v255, 0" ::: "v255");
    asm volatile("v_accvgpr_write_b32 a23, 0" ::: "a23");
    pad[threadIdx.x] = (int)ticks;
    const long long t0 = wall_clock64();
    while (wall_clock64() - t0 < ticks) __builtin_amdgcn_s_sleep(8);
    if (ticks < 0) sink[threadIdx.x] = pad[255 - threadIdx.x];
}
int devpack_spin(void *stream, int micros, int channels)
{
    if (micros <= 0) return DASP_OK;
    if (channels > 0) hipLaunchKernelGGL(k_spin_fat, dim3(channels), dim3(256), 0, static_cast<hipStream_t>(stream), (long long)micros * 100, (int *)nullptr);
    else hipLaunchKernelGGL(k_spin_us, dim3(1), dim3(64), 0, static_cast<hipStream_t>(stream), (long long)micros * 100);
    HIP_TRYP(hipGetLastError());
    return DASP_OK;
}

// copy a packed array back to the host (tests; serialising a device-built plan)
int download_array(Plan &p, const char *name, void *dst, size_t bytes)
{
    if (!p.dev || !p.dev->arena) { set_error("plan not on the device"); return DASP_ERR_STATE; }
    const ArenaMap &mp = p.dev->map;
    const size_t vb = (size_t)p.geo.vbytes;
    if (p.two_phase) {      // the tile streams of a two-phase plan (tp_xs: what phase 1 wrote in the last product)
        const TpDev &q = p.dev->tp;
        const size_t S = p.tp.segments;
        struct { const char *n; const void *ptr; size_t len; } tt[] = {
            {"tp_lcol", q.lcol, S * kTpSeg * 2}, {"tp_dst", q.dst, S * 4}, {"tp_val", q.val, S * kTpSeg * vb}, {"tp_lrow", q.lrow, S * kTpSeg * 2}, {"tp_xs", q.xs, S * kTpSeg * vb}};
        for (auto &t : tt)
            if (std::strcmp(t.n, name) == 0) {
                if (bytes != t.len) { set_error(std::string("size mismatch for ") + name); return DASP_ERR_ARG; }
                if (t.len) HIP_TRYP(hipMemcpy(dst, t.ptr, t.len, hipMemcpyDeviceToHost));
                return DASP_OK;
            }
        set_error(std::string("unknown device array of a two-phase plan: ") + name);
        return DASP_ERR_ARG;
    }
    struct { const char *n; size_t off, len; } tab[] = {
        {"long_val", mp.long_val, p.cnt_long * vb}, {"long_cid", mp.long_cid, p.cnt_long * 4}, {"long_cid16", mp.long_cid16, p.cnt_long * 2}, {"long_base", mp.long_base, p.cnt_long_chunks * 4},
        {"med_val", mp.med_val, p.cnt_reg * vb}, {"med_cid", mp.med_cid, p.cid16 ? 0 : p.cnt_reg * 4},
        {"med_cid16", mp.med_cid16, p.cid16 ? (p.cnt_reg - p.cnt_reg8) * 2 : 0}, {"med_cid8", mp.med_cid8, p.cnt_reg8}, {"med_base", mp.med_base, p.cid16 ? (size_t)p.med_ptr.back() * 4 : 0},
        {"irr_val", mp.irr_val, p.cnt_irr * vb}, {"irr_cid", mp.irr_cid, p.cnt_irr * 4},
        {"short_val", mp.short_val, p.cnt_short * vb}, {"short_cid", mp.short_cid, p.cnt_short * 4},
        {"rt_val", mp.rt_val, p.cnt_rt * vb}, {"rt_cid", mp.rt_cid, p.cnt_rt * 4},
    };
    for (auto &t : tab)
        if (std::strcmp(t.n, name) == 0) {
            if (bytes != t.len) { set_error(std::string("size mismatch for ") + name); return DASP_ERR_ARG; }
            if (t.len) HIP_TRYP(hipMemcpy(dst, static_cast<char *>(p.dev->arena) + t.off, t.len, hipMemcpyDeviceToHost));
            return DASP_OK;
        }
    set_error(std::string("unknown device array: ") + name);
    return DASP_ERR_ARG;
}

}  // namespace dasp

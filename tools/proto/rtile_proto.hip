// tools/proto/rtile_proto.hip -- prototype, not part of the library (one shared row order for the column panels).
// A column panel's rows of at most T nonzeros, kept in the PARENT's slot order in tiles of 64 consecutive slots: one wave per tile reads the
// tile's entries as a stream (value, column), parks the products in LDS, lane r sums row r's products in their CSR order and the wave stores
// one complete 128-byte line of the panel's partial y.  Build: hipcc -O3 --offload-arch=gfx950 -shared -fPIC -o librtile.so rtile_proto.hip
#include <hip/hip_runtime.h>

template <int T>
__global__ __launch_bounds__(256) void rtile_kernel(const _Float16 *val, const int *cid, const unsigned short *rowstart, const unsigned long long *mask,
                                                    const int *tile_ptr, int n_tiles, const _Float16 *x, _Float16 *ypart)
{
    __shared__ float prod[4][64 * T];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, t = blockIdx.x * 4 + wave;
    if (t >= n_tiles) return;
    const int e0 = tile_ptr[t], n = tile_ptr[t + 1] - e0;
    float *pw = prod[wave];
#pragma unroll 4
    for (int i = lane; i < n; i += 64) pw[i] = (float)__builtin_nontemporal_load(val + e0 + i) * (float)x[__builtin_nontemporal_load(cid + e0 + i)];
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
    __builtin_amdgcn_wave_barrier();
    const int s = rowstart[(size_t)t * 64 + lane];
    int e = __shfl_down(s, 1);
    if (lane == 63) e = n;
    float sum = 0;
    for (int j = s; j < e; ++j) sum += pw[j];
    if ((mask[t] >> lane) & 1) ypart[(size_t)t * 64 + lane] = (_Float16)sum;
}

// variant B: the wave's loads in batches of four chunks (clamped, not branched), tile bounds through scalar loads, 8 waves per SIMD asked for;
// R = 1 / 2: tiles of 64 / 128 slots (lane owns rows lane and lane + 64: two lines per wave)
template <int T, int R>
__global__ __launch_bounds__(256, 8) void rtile_kernel_b(const _Float16 *val, const int *cid, const unsigned short *rowstart, const unsigned long long *mask,
                                                         const int *tile_ptr, int n_tiles, const _Float16 *x, _Float16 *ypart)
{
    __shared__ float prod[4][64 * T * R];
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)), lane = threadIdx.x & 63, t = blockIdx.x * 4 + wave;      // t: unit of R tiles
    if (t * R >= n_tiles) return;
    const int tl = min(t * R + R, n_tiles);
    const int e0 = tile_ptr[t * R], n = tile_ptr[tl] - e0;
    float *pw = prod[wave];
    int s[R];
#pragma unroll
    for (int r = 0; r < R; ++r) s[r] = t * R + r < n_tiles ? rowstart[(size_t)(t * R + r) * 64 + lane] : 0;
    for (int base = 0; base < n; base += 256) {
        float v[4]; int c[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int i = min(base + u * 64 + lane, n - 1);
            v[u] = (float)__builtin_nontemporal_load(val + e0 + i); c[u] = __builtin_nontemporal_load(cid + e0 + i);
        }
        float xv[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) xv[u] = (float)x[c[u]];
#pragma unroll
        for (int u = 0; u < 4; ++u) { const int i = base + u * 64 + lane; if (i < n) pw[i] = v[u] * xv[u]; }
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int r = 0; r < R; ++r) {
        if (t * R + r >= n_tiles) break;
        const int off = r == 0 ? 0 : tile_ptr[t * R + r] - e0;           // rowstart is relative to its own tile
        const int nn = (r + 1 < R && t * R + r + 1 < n_tiles ? tile_ptr[t * R + r + 1] : tile_ptr[tl]) - e0 - off;
        int e = __shfl_down(s[r], 1);
        if (lane == 63) e = nn;
        float sum = 0;
        for (int j = s[r]; j < e; ++j) sum += pw[off + j];
        if ((mask[t * R + r] >> lane) & 1) ypart[(size_t)(t * R + r) * 64 + lane] = (_Float16)sum;
    }
}

// variant C: R consecutive tiles are one unit of a wave: their entries are contiguous, so one product loop, then R row-sum rounds
template <int T, int R>
__global__ __launch_bounds__(256) void rtile_kernel_c(const _Float16 *val, const int *cid, const unsigned short *rowstart, const unsigned long long *mask,
                                                      const int *tile_ptr, int n_tiles, const _Float16 *x, _Float16 *ypart)
{
    __shared__ float prod[4][64 * T * R];
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)), lane = threadIdx.x & 63, t0 = (blockIdx.x * 4 + wave) * R;
    if (t0 >= n_tiles) return;
    const int nt = min(R, n_tiles - t0);
    const int e0 = tile_ptr[t0], n = tile_ptr[t0 + nt] - e0;
    float *pw = prod[wave];
#pragma unroll 4
    for (int i = lane; i < n; i += 64) pw[i] = (float)__builtin_nontemporal_load(val + e0 + i) * (float)x[__builtin_nontemporal_load(cid + e0 + i)];
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
    __builtin_amdgcn_wave_barrier();
    for (int r = 0; r < nt; ++r) {
        const int off = tile_ptr[t0 + r] - e0, nn = tile_ptr[t0 + r + 1] - e0 - off;
        const int s = rowstart[(size_t)(t0 + r) * 64 + lane];
        int e = __shfl_down(s, 1);
        if (lane == 63) e = nn;
        float sum = 0;
        for (int j = s; j < e; ++j) sum += pw[off + j];
        if ((mask[t0 + r] >> lane) & 1) ypart[(size_t)(t0 + r) * 64 + lane] = (_Float16)sum;
    }
}

// y[i] = sum_k part[k * stride + i], 8 values per thread
__global__ __launch_bounds__(256) void rtile_sum_kernel(const _Float16 *part, size_t stride, int P, _Float16 *y, int n8)
{
    typedef _Float16 h8 __attribute__((ext_vector_type(8)));
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n8) return;
    float acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    for (int k = 0; k < P; ++k) {
        const h8 v = __builtin_nontemporal_load(reinterpret_cast<const h8 *>(part + k * stride) + i);
#pragma unroll
        for (int j = 0; j < 8; ++j) acc[j] += (float)v[j];
    }
    h8 o;
#pragma unroll
    for (int j = 0; j < 8; ++j) o[j] = (_Float16)acc[j];
    __builtin_nontemporal_store(o, reinterpret_cast<h8 *>(y) + i);
}

extern "C" int rtile_launch_b(int T, int R, const void *val, const void *cid, const void *rowstart, const void *mask, const void *tile_ptr, int n_tiles, const void *x, void *ypart, void *stream)
{
    const dim3 grid(((n_tiles + R - 1) / R + 3) / 4), block(256);
    auto s = static_cast<hipStream_t>(stream);
#define GO(TT, RR) hipLaunchKernelGGL((rtile_kernel_b<TT, RR>), grid, block, 0, s, (const _Float16 *)val, (const int *)cid, (const unsigned short *)rowstart, (const unsigned long long *)mask, (const int *)tile_ptr, n_tiles, (const _Float16 *)x, (_Float16 *)ypart)
    if (R == 1) { if (T <= 8) GO(8, 1); else if (T <= 16) GO(16, 1); else return -1; }
    else if (R == 2) { if (T <= 8) GO(8, 2); else if (T <= 16) GO(16, 2); else return -1; }
#undef GO
#define GO(TT, RR) hipLaunchKernelGGL((rtile_kernel_c<TT, RR>), dim3(((n_tiles + RR - 1) / RR + 3) / 4), block, 0, s, (const _Float16 *)val, (const int *)cid, (const unsigned short *)rowstart, (const unsigned long long *)mask, (const int *)tile_ptr, n_tiles, (const _Float16 *)x, (_Float16 *)ypart)
    else if (R == -2) { if (T <= 8) GO(8, 2); else if (T <= 16) GO(16, 2); else return -1; }
    else if (R == -4) { if (T <= 8) GO(8, 4); else if (T <= 16) GO(16, 4); else return -1; }
    else if (R == -8) { if (T <= 8) GO(8, 8); else return -1; }
    else return -1;
#undef GO
    return (int)hipGetLastError();
}
extern "C" int rtile_launch(int T, const void *val, const void *cid, const void *rowstart, const void *mask, const void *tile_ptr, int n_tiles, const void *x, void *ypart, void *stream)
{
    const dim3 grid((n_tiles + 3) / 4), block(256);
    auto s = static_cast<hipStream_t>(stream);
#define GO(TT) hipLaunchKernelGGL(rtile_kernel<TT>, grid, block, 0, s, (const _Float16 *)val, (const int *)cid, (const unsigned short *)rowstart, (const unsigned long long *)mask, (const int *)tile_ptr, n_tiles, (const _Float16 *)x, (_Float16 *)ypart)
    if (T <= 4) GO(4); else if (T <= 8) GO(8); else if (T <= 16) GO(16); else if (T <= 32) GO(32); else return -1;
#undef GO
    return (int)hipGetLastError();
}
extern "C" int rtile_sum(const void *part, size_t stride, int P, void *y, int n, void *stream)
{
    hipLaunchKernelGGL(rtile_sum_kernel, dim3((n / 8 + 255) / 256), dim3(256), 0, static_cast<hipStream_t>(stream), (const _Float16 *)part, stride, P, (_Float16 *)y, n / 8);
    return (int)hipGetLastError();
}

// tools/micro/pb_kernels.hip -- propagation blocking as an EXPERIMENT for the gather-bound graph matrices (not part of the library): y = A x in two
// streaming phases instead of one gather per nonzero.  tools/micro/pb_probe.py builds the layout and drives it.
//   phase 1, one workgroup per block of C columns: the block's x slice goes to LDS; its nonzeros -- stored sorted by the row bin they feed --
//            are read in order (value f16, column offset u16), multiplied, and the products written to the bin-major buffer; a tile
//            (column block x row bin) is contiguous in both orders, so a table of (source offset, destination offset, length) per tile is all
//            the addressing there is
//   phase 2, one workgroup per bin of B rows: f32 accumulators in LDS, the bin's products (f32) and their static row offsets (u16) streamed in,
//            ds_add_f32, then y (f16) written out coalesced
// hipcc --offload-arch=gfx950 -O3 -shared -fPIC tools/micro/pb_kernels.hip -o build/micro/libpb.so
#include <hip/hip_runtime.h>
#include <cstdint>

struct Tile { int src, dst, len; };

__global__ __launch_bounds__(256) void pb_phase1(const _Float16 *val, const uint16_t *lcol, const Tile *tiles, const int *tile_ptr, const _Float16 *x, int C, int n,
                                                 float *contrib)
{
    extern __shared__ _Float16 xs[];
    const int cb = blockIdx.x;
    const int c0 = cb * C, cn = min(C, n - c0);
    for (int i = threadIdx.x; i < cn; i += 256) xs[i] = x[c0 + i];
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int t = tile_ptr[cb] + wave; t < tile_ptr[cb + 1]; t += 4) {
        const Tile T = tiles[t];
        for (int k = lane; k < T.len; k += 64)
            contrib[T.dst + k] = (float)val[T.src + k] * (float)xs[lcol[T.src + k]];
    }
}

// flat variant of phase 1: one thread per nonzero, the tile of a nonzero found from a per-nonzero destination array (4 more bytes per nonzero, no per-tile loop)
__global__ __launch_bounds__(256) void pb_phase1_flat(const _Float16 *val, const uint16_t *lcol, const int *dst, const int *cb_ptr, const _Float16 *x, int C, int n,
                                                      float *contrib)
{
    extern __shared__ _Float16 xs[];
    const int cb = blockIdx.x;
    const int c0 = cb * C, cn = min(C, n - c0);
    for (int i = threadIdx.x; i < cn; i += 256) xs[i] = x[c0 + i];
    __syncthreads();
    for (int k = cb_ptr[cb] + threadIdx.x; k < cb_ptr[cb + 1]; k += 256) contrib[dst[k]] = (float)val[k] * (float)xs[lcol[k]];
}

__global__ __launch_bounds__(1024) void pb_phase2(const float *contrib, const uint16_t *lrow, const int *bin_ptr, int B, int m, _Float16 *y)
{
    extern __shared__ float acc[];
    const int b = blockIdx.x;
    const int r0 = b * B, rn = min(B, m - r0);
    for (int i = threadIdx.x; i < rn; i += blockDim.x) acc[i] = 0.f;
    __syncthreads();
    const int k0 = bin_ptr[b], k1 = bin_ptr[b + 1];
    for (int k = k0 + threadIdx.x; k < k1; k += blockDim.x) atomicAdd(&acc[lrow[k]], contrib[k]);
    __syncthreads();
    for (int i = threadIdx.x; i < rn; i += blockDim.x) y[r0 + i] = (_Float16)acc[i];
}

extern "C" {
// times `iters` SpMVs (both phases) with events; flat != 0: the flat phase 1.  Returns ms per SpMV, < 0 on error.
float pb_time(const void *val, const void *lcol, const void *tiles, const void *tile_ptr, const void *dst, const void *cb_ptr, const void *x, int C, int n, int ncb,
              void *contrib, const void *lrow, const void *bin_ptr, int B, int m, int nb, void *y, int flat, int warm, int iters, float *ms1, float *ms2)
{
    hipEvent_t e0, e1, e2;
    hipEventCreate(&e0); hipEventCreate(&e1); hipEventCreate(&e2);
    float t1 = 0, t2 = 0;
    for (int it = 0; it < warm + iters; ++it) {
        hipEventRecord(e0, 0);
        if (flat) hipLaunchKernelGGL(pb_phase1_flat, dim3(ncb), dim3(256), C * 2, 0, (const _Float16 *)val, (const uint16_t *)lcol, (const int *)dst, (const int *)cb_ptr,
                                     (const _Float16 *)x, C, n, (float *)contrib);
        else hipLaunchKernelGGL(pb_phase1, dim3(ncb), dim3(256), C * 2, 0, (const _Float16 *)val, (const uint16_t *)lcol, (const Tile *)tiles, (const int *)tile_ptr,
                                (const _Float16 *)x, C, n, (float *)contrib);
        hipEventRecord(e1, 0);
        hipLaunchKernelGGL(pb_phase2, dim3(nb), dim3(1024), B * 4, 0, (const float *)contrib, (const uint16_t *)lrow, (const int *)bin_ptr, B, m, (_Float16 *)y);
        hipEventRecord(e2, 0);
        if (hipEventSynchronize(e2) != hipSuccess) return -1.f;
        if (it >= warm) { float a, b; hipEventElapsedTime(&a, e0, e1); hipEventElapsedTime(&b, e1, e2); t1 += a; t2 += b; }
    }
    if (hipGetLastError() != hipSuccess) return -1.f;
    *ms1 = t1 / iters; *ms2 = t2 / iters;
    return (t1 + t2) / iters;
}
}

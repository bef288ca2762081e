// tools/micro/rw_mix.hip -- profiles/r03_placement.md: is it the y WRITES against the arena READS?  Four 2.1-GB read buffers ("arenas": three streams 8 : 2 : 1 bytes)
// x eight 9.6-MB write buffers ("y": 128 B per block at the block's end), every combination timed; plus the read-only time per arena.
// hipcc --offload-arch=gfx950 -O3 tools/micro/rw_mix.hip -o build/micro/rw_mix
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define CK(e) do { hipError_t r_ = (e); if (r_ != hipSuccess) { printf("%s: %s\n", #e, hipGetErrorString(r_)); return 1; } } while (0)

template <bool WRITE>
__global__ __launch_bounds__(256) void mix(const double *v, const unsigned short *c16, const unsigned char *c8, size_t nblk, double *y)
{
    const size_t b = (size_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (b >= nblk) return;
    double acc = 0;
    for (int s = 0; s < 20; ++s) {
        const size_t e = (b * 20 + s) * 128 + lane * 2;
        const double2 x = *reinterpret_cast<const double2 *>(v + e);
        const unsigned id = *reinterpret_cast<const unsigned *>(c16 + e);
        const unsigned short n8 = *reinterpret_cast<const unsigned short *>(c8 + e);
        acc += x.x + x.y + (double)(id & 7) + (double)(n8 & 3);
    }
    for (int o = 32; o >= 16; o >>= 1) acc += __shfl_xor(acc, o);   // every lane's loads are needed
    if (WRITE) { if (lane < 16) y[b * 16 + lane] = acc; }          // 16 rows of the block: 128 B
    else if (acc == 12345.678) y[b] = acc;
}

int main()
{
    const size_t nblk = 75000;
    const size_t vb = nblk * 20 * 1024, cb = nblk * 20 * 256, nb8 = nblk * 20 * 128, total = vb + cb + nb8;
    const int NA = 4, NY = 8;
    std::vector<char *> A(NA); std::vector<double *> Y(NY);
    for (int i = 0; i < NA; ++i) { CK(hipMalloc((void **)&A[i], total)); CK(hipMemset(A[i], 1, total)); }
    for (int j = 0; j < NY; ++j) { CK(hipMalloc((void **)&Y[j], nblk * 16 * 8)); CK(hipMemset(Y[j], 0, nblk * 16 * 8)); }
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    auto run = [&](int i, int j, bool wr, float *ms) -> int {
        const double *v = (const double *)A[i]; const unsigned short *c16 = (const unsigned short *)(A[i] + vb); const unsigned char *c8 = (const unsigned char *)(A[i] + vb + cb);
        const dim3 g((unsigned)((nblk + 3) / 4));
        for (int k = 0; k < 5; ++k) { if (wr) hipLaunchKernelGGL(mix<true>, g, dim3(256), 0, 0, v, c16, c8, nblk, Y[j]); else hipLaunchKernelGGL(mix<false>, g, dim3(256), 0, 0, v, c16, c8, nblk, Y[j]); }
        CK(hipEventRecord(e0, 0));
        for (int k = 0; k < 100; ++k) { if (wr) hipLaunchKernelGGL(mix<true>, g, dim3(256), 0, 0, v, c16, c8, nblk, Y[j]); else hipLaunchKernelGGL(mix<false>, g, dim3(256), 0, 0, v, c16, c8, nblk, Y[j]); }
        CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
        CK(hipEventElapsedTime(ms, e0, e1)); *ms /= 100;
        return 0;
    };
    for (int i = 0; i < NA; ++i) {
        float ms; if (run(i, 0, false, &ms)) return 1;
        printf("arena %d read only %.4f ms | with y buffer 0..%d:", i, ms, NY - 1);
        for (int j = 0; j < NY; ++j) { if (run(i, j, true, &ms)) return 1; printf(" %.4f", ms); }
        printf("\n");
    }
    return 0;
}

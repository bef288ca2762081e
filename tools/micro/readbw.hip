// tools/micro/readbw.hip -- what a pure streaming READ reaches on this GPU, by load width and cache policy
// (the ceiling the SpMV kernel's actual byte rate is compared with in DESIGN.md).  hipcc --offload-arch=gfx950 -O3
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double f64x2 __attribute__((ext_vector_type(2)));
template <class V, bool NT>
__global__ void rd(const V *p, size_t n, double *out)
{
    double s = 0;
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        V v = NT ? __builtin_nontemporal_load(p + i) : p[i];
        if constexpr (sizeof(V) == 16) s += v[0] + v[1]; else s += v;
    }
    if (s == 12345.678) out[0] = s;   // keep the loads alive
}
template <class V, bool NT> void run(const char *name, const void *buf, size_t bytes, double *out, int grid)
{
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    const size_t n = bytes / sizeof(V);
    for (int i = 0; i < 3; ++i) hipLaunchKernelGGL((rd<V, NT>), dim3(grid), dim3(256), 0, 0, (const V *)buf, n, out);
    hipEventRecord(a);
    for (int i = 0; i < 20; ++i) hipLaunchKernelGGL((rd<V, NT>), dim3(grid), dim3(256), 0, 0, (const V *)buf, n, out);
    hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    std::printf("%-28s grid %6d : %.3f ms  %.1f GB/s\n", name, grid, ms / 20, bytes / (ms / 20 * 1e6));
}
int main()
{
    const size_t bytes = (size_t)3 << 30;
    void *buf; double *out;
    hipMalloc(&buf, bytes); hipMalloc(&out, 8); hipMemset(buf, 0, bytes);
    for (int grid : {2048, 8192, 65536}) {
        run<double, false>("8 B/lane default", buf, bytes, out, grid);
        run<double, true>("8 B/lane nt", buf, bytes, out, grid);
        run<f64x2, false>("16 B/lane default", buf, bytes, out, grid);
        run<f64x2, true>("16 B/lane nt", buf, bytes, out, grid);
    }
    return 0;
}

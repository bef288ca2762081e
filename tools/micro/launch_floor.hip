// tools/micro/launch_floor.hip -- the period of back-to-back launches on one stream: an empty kernel, one with a 512-byte argument block it reads, one that stores a word.
// hipcc -O3 --offload-arch=gfx950 -o launch_floor launch_floor.hip
#include <hip/hip_runtime.h>
#include <cstdio>
struct Big { long long v[64]; };
__global__ void k_empty() {}
__global__ void k_args(Big b, long long *out) { if (b.v[threadIdx.x & 63] == 0x7fffffffffffffffll) *out = 1; }
__global__ void k_store(long long *out) { if (threadIdx.x == 0) out[blockIdx.x] = blockIdx.x; }
template <class F> static double period(F launch, int n)
{
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    for (int i = 0; i < 200; ++i) launch();
    hipDeviceSynchronize();
    hipEventRecord(a, 0);
    for (int i = 0; i < n; ++i) launch();
    hipEventRecord(b, 0); hipEventSynchronize(b);
    float ms = 0; hipEventElapsedTime(&ms, a, b);
    return ms * 1e3 / n;
}
int main()
{
    long long *out; hipMalloc(&out, 1 << 20);
    Big big{}; 
    printf("empty kernel, 1 workgroup:        %.2f us per launch\n", period([&] { hipLaunchKernelGGL(k_empty, dim3(1), dim3(64), 0, 0); }, 2000));
    printf("empty kernel, 1024 workgroups:    %.2f us\n", period([&] { hipLaunchKernelGGL(k_empty, dim3(1024), dim3(256), 0, 0); }, 2000));
    printf("512-byte arguments read, 1 wg:    %.2f us\n", period([&] { hipLaunchKernelGGL(k_args, dim3(1), dim3(64), 0, 0, big, out); }, 2000));
    printf("one store per workgroup, 1024 wg: %.2f us\n", period([&] { hipLaunchKernelGGL(k_store, dim3(1024), dim3(256), 0, 0, out); }, 2000));
    return 0;
}

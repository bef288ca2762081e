// tools/micro/launch_floor2.hip -- r5: what the 528-byte by-value DevArgs block costs a small launch, and what would replace it.
// Back-to-back launches on one stream of a kernel shaped like dasp_spmv_win1_kernel's cop20k_A launch (212 workgroups x 1024 threads), whose every
// wave reads its arguments with SCALAR loads (as the product does) and stores nothing:
//   by value   : the whole block in the kernarg segment (today)
//   pointer    : kernarg = one pointer to a device-resident copy (+ x, y), scalar-loaded through the constant address space
//   hot / cold : 64 hot bytes by value + a pointer to the cold rest
// hipcc -O3 --offload-arch=gfx950 -o launch_floor2 launch_floor2.hip
#include <hip/hip_runtime.h>
#include <cstdio>
struct Big { long long v[66]; };          // 528 bytes
struct Hot { long long v[8]; };           // 64 bytes
typedef const __attribute__((address_space(4))) long long *cptr;
__device__ __forceinline__ long long fold(const long long *v, int n, int w) { long long s = 0; for (int i = 0; i < n; ++i) s += v[(i * 7 + w) % n]; return s; }
__global__ __launch_bounds__(1024) void k_value(Big b, long long *out)
{
    const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    long long s = 0;
#pragma unroll
    for (int i = 0; i < 66; ++i) s ^= b.v[i] + w;
    if (s == 0x7fffffffffffffffll) *out = s;
}
__global__ __launch_bounds__(1024) void k_pointer(const Big *bp, long long *out)
{
    const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    cptr p = (cptr)(uintptr_t)bp;
    long long s = 0;
#pragma unroll
    for (int i = 0; i < 66; ++i) s ^= p[i] + w;
    if (s == 0x7fffffffffffffffll) *out = s;
}
__global__ __launch_bounds__(1024) void k_hotcold(Hot h, const Big *bp, long long *out)
{
    const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    cptr p = (cptr)(uintptr_t)bp;
    long long s = 0;
#pragma unroll
    for (int i = 0; i < 8; ++i) s ^= h.v[i] + w;
#pragma unroll
    for (int i = 8; i < 66; ++i) s ^= p[i] + w;
    if (s == 0x7fffffffffffffffll) *out = s;
}
__global__ __launch_bounds__(1024) void k_small(Hot h, long long *out)
{
    const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    long long s = 0;
#pragma unroll
    for (int i = 0; i < 8; ++i) s ^= h.v[i] + w;
    if (s == 0x7fffffffffffffffll) *out = s;
}
__global__ __launch_bounds__(1024) void k_empty() {}
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
    Big big{}; Hot hot{};
    for (int i = 0; i < 66; ++i) big.v[i] = i + 1;
    for (int i = 0; i < 8; ++i) hot.v[i] = i + 1;
    Big *dbig; hipMalloc(&dbig, sizeof(Big)); hipMemcpy(dbig, &big, sizeof big, hipMemcpyHostToDevice);
    for (int grid : {1, 212, 1024}) {
        printf("grid %4d x 1024 threads:  empty %.2f us | 64 B by value %.2f | 528 B by value %.2f | pointer to 528 B on the device %.2f | 64 B hot + pointer %.2f\n", grid,
               period([&] { hipLaunchKernelGGL(k_empty, dim3(grid), dim3(1024), 0, 0); }, 3000),
               period([&] { hipLaunchKernelGGL(k_small, dim3(grid), dim3(1024), 0, 0, hot, out); }, 3000),
               period([&] { hipLaunchKernelGGL(k_value, dim3(grid), dim3(1024), 0, 0, big, out); }, 3000),
               period([&] { hipLaunchKernelGGL(k_pointer, dim3(grid), dim3(1024), 0, 0, dbig, out); }, 3000),
               period([&] { hipLaunchKernelGGL(k_hotcold, dim3(grid), dim3(1024), 0, 0, hot, dbig, out); }, 3000));
    }
    // the same inside a graph of 50 launches (no host launch cost)
    return 0;
}

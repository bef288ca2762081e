// tools/micro/gather_tags.hip -- how many L1 tag lookups (TCP_TOTAL_CACHE_ACCESSES) one 64-lane GATHER instruction costs, by element
// size and by how the lanes' addresses group.  Run under: rocprofv3 --kernel-trace --pmc TCP_TOTAL_CACHE_ACCESSES_sum SQ_INSTS_VMEM_RD --
// hipcc --offload-arch=gfx950 -O3 tools/micro/gather_tags.hip -o gather_tags
#include <hip/hip_runtime.h>
#include <cstdio>
// pattern p: the element index lane l reads = base(wave, iter) + off_p(l)
//  0: all 64 lanes the same element          1: quads share an element, quads 1 KB apart      2: quads share an element, quads 16 B apart
//  3: lanes of a quad 10 B .. apart (adjacent elements 5 apart), quads 1 KB apart            4: 16-lane groups share an element, groups 1 KB apart
//  5: every lane its own 128-B line          6: lanes contiguous (coalesced stream)          7: quad lanes = 2 distinct elements 5 apart, quads 1 KB apart
template <class T, int P>
__global__ void g(const T *x, size_t n, double *out, int iters)
{
    const int l = threadIdx.x & 63;
    const size_t wave = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    size_t off;
    const int q = l >> 2, r = l & 3, g16 = l >> 4;
    if (P == 0) off = 0;
    else if (P == 1) off = (size_t)q * (1024 / sizeof(T));
    else if (P == 2) off = (size_t)q * (16 / sizeof(T));
    else if (P == 3) off = (size_t)q * (1024 / sizeof(T)) + 5 * r;
    else if (P == 4) off = (size_t)g16 * (1024 / sizeof(T));
    else if (P == 5) off = (size_t)l * (128 / sizeof(T));
    else if (P == 6) off = l;
    else off = (size_t)q * (1024 / sizeof(T)) + 5 * (r >> 1);
    double s = 0;
    for (int it = 0; it < iters; ++it) {
        const size_t base = ((wave * 131 + (size_t)it * 7919) * 4096) % (n - 65536);
        s += (double)x[base + off];
    }
    if (s == 12345.678) out[0] = s;
}
template <class T, int P> void run(const T *x, size_t n, double *out)
{
    hipLaunchKernelGGL((g<T, P>), dim3(4096), dim3(256), 0, 0, x, n, out, 64);
}
int main()
{
    const size_t bytes = (size_t)64 << 20;
    void *buf; double *out;
    if (hipMalloc(&buf, bytes) != hipSuccess || hipMalloc(&out, 8) != hipSuccess) return 1;
    (void)hipMemset(buf, 0, bytes);
#define ALL(T) run<T, 0>((const T *)buf, bytes / sizeof(T), out); run<T, 1>((const T *)buf, bytes / sizeof(T), out); run<T, 2>((const T *)buf, bytes / sizeof(T), out); \
               run<T, 3>((const T *)buf, bytes / sizeof(T), out); run<T, 4>((const T *)buf, bytes / sizeof(T), out); run<T, 5>((const T *)buf, bytes / sizeof(T), out); \
               run<T, 6>((const T *)buf, bytes / sizeof(T), out); run<T, 7>((const T *)buf, bytes / sizeof(T), out);
    ALL(_Float16) ALL(float) ALL(double)
    (void)hipDeviceSynchronize();
    std::printf("done\n");
    return 0;
}

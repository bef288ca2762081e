// tools/micro/sgather.hip -- can the SCALAR path (s_load: SQC -> L2, its own miss handling) carry part of an L2-resident gather that the vector L1's miss queue
// bounds?  idx: N random positions in an x of X floats (X * 4 B = 2.75 MB: an XCD's L2 holds it).  Per wave and step of 64 positions: all 64 through the
// vector path, or S of them through v_readlane + s_load_dword + v_cndmask and the rest through the vector path.
// hipcc -O3 --offload-arch=gfx950 -o sgather sgather.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

template <int S>      // S of every 64 gathers through the scalar path (0: vector only; 64: scalar only)
__global__ __launch_bounds__(256) void gather(const int *__restrict__ idx, size_t nsteps, const float *__restrict__ x, float *out)
{
    const int lane = threadIdx.x & 63;
    const size_t wave = __builtin_amdgcn_readfirstlane((int)(((size_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6));
    const size_t nw = ((size_t)gridDim.x * blockDim.x) >> 6;
    float acc = 0;
    for (size_t i = wave; i < nsteps; i += 2 * nw) {
        const size_t j = i + nw < nsteps ? i + nw : i;
        const int c0 = __builtin_nontemporal_load(idx + i * 64 + lane), c1 = __builtin_nontemporal_load(idx + j * 64 + lane);      // two steps in flight
        float g0 = 0, g1 = 0;
        if (S < 64) { if (lane >= S) { g0 = x[c0]; g1 = x[c1]; } }
        if constexpr (S > 0) {
            float s0 = 0, s1 = 0;
#pragma unroll
            for (int l = 0; l < S; ++l) {
                const int a = __builtin_amdgcn_readlane(c0, l), b = __builtin_amdgcn_readlane(c1, l);
                const float va = x[a], vb = x[b];                        // uniform addresses: s_load_dword
                s0 = lane == l ? va : s0;                                 // (v_cndmask with a uniform source: what v_writelane would do)
                s1 = lane == l ? vb : s1;
            }
            g0 += s0; g1 += s1;
        }
        acc += g0 + g1;
    }
    if (acc == 12345.678f) out[0] = acc;
}

template <class F> static double ms_of(F launch)
{
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    for (int i = 0; i < 2; ++i) launch();
    hipEventRecord(a, 0);
    for (int i = 0; i < 5; ++i) launch();
    hipEventRecord(b, 0); hipEventSynchronize(b);
    float ms = 0; hipEventElapsedTime(&ms, a, b);
    return ms / 5;
}
int main(int argc, char **argv)
{
    const size_t N = (size_t)64 << 20;
    const int X = argc > 1 ? atoi(argv[1]) : 720896;            // floats: 2.75 MiB
    std::vector<int> h(N);
    unsigned long long s = 88172645463325252ull;
    for (size_t i = 0; i < N; ++i) { s ^= s << 13; s ^= s >> 7; s ^= s << 17; h[i] = (int)(s % (unsigned long long)X); }
    int *idx; float *x, *out;
    hipMalloc(&idx, N * 4); hipMalloc(&x, (size_t)X * 4); hipMalloc(&out, 64);
    hipMemcpy(idx, h.data(), N * 4, hipMemcpyHostToDevice); hipMemset(x, 0, (size_t)X * 4);
    const int grid = 256 * (argc > 2 ? atoi(argv[2]) : 8);      // resident workgroups (of 4 waves) per CU
    printf("x of %d floats (%.2f MiB), %zu M gathers, %d workgroups per CU\n", X, X * 4.0 / (1 << 20), N >> 20, grid / 256);
#define RUN(SV) { const double ms = ms_of([&] { hipLaunchKernelGGL(gather<SV>, dim3(grid), dim3(256), 0, 0, idx, N / 64, x, out); }); \
                  printf("%2d of 64 through the scalar path: %.3f ms = %.1f G gathers/s\n", SV, ms, N / ms / 1e6); }
    RUN(0) RUN(4) RUN(8) RUN(16) RUN(32) RUN(64) RUN(0)
    return 0;
}

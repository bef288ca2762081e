// tools/micro/memops.cpp -- does this device / runtime support stream memory operations (hipStreamWriteValue64 / hipStreamWaitValue64), on which memory,
// and what does a cross-stream hand-off cost through them vs through an event?   hipcc -O2 memops.cpp -o memops
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <chrono>
#include <cstdio>
__global__ void spin(int n, int *out) { int s = 0; for (int i = 0; i < n; ++i) s += i; if (s == 42) *out = s; }
int main()
{
    int can = -1;
    printf("attr rc=%d can=%d\n", (int)hipDeviceGetAttribute(&can, hipDeviceAttributeCanUseStreamWaitValue, 0), can);
    void *sig = nullptr, *plain = nullptr; int *dummy;
    printf("signal malloc rc=%d\n", (int)hipExtMallocWithFlags(&sig, 16, hipMallocSignalMemory));
    printf("plain malloc rc=%d\n", (int)hipMalloc(&plain, 16));
    hipMalloc(&dummy, 4);
    hipMemset(plain, 0, 16); if (sig) hipMemset(sig, 0, 16);
    hipStream_t a, b; hipStreamCreateWithFlags(&a, hipStreamNonBlocking); hipStreamCreateWithFlags(&b, hipStreamNonBlocking);
    for (void *p : {sig, plain}) {
        if (!p) continue;
        hipError_t w = hipStreamWriteValue64(a, p, 1, 0);
        hipError_t q = hipStreamWaitValue64(b, p, 1, hipStreamWaitValueGte, 0xFFFFFFFFFFFFFFFFull);
        hipError_t s1 = hipStreamSynchronize(a), s2 = hipStreamSynchronize(b);
        printf("%s: write rc=%d wait rc=%d sync %d %d\n", p == sig ? "signal" : "plain", (int)w, (int)q, (int)s1, (int)s2);
        (void)hipGetLastError();
    }
    // hand-off latency: kernel on a -> [hand-off] -> kernel on b -> [hand-off] -> kernel on a ...  (ping-pong), n rounds
    hipEvent_t ea, eb, t0, t1; hipEventCreateWithFlags(&ea, hipEventDisableTiming); hipEventCreateWithFlags(&eb, hipEventDisableTiming);
    hipEventCreate(&t0); hipEventCreate(&t1);
    const int n = 200;
    for (int mode = 0; mode < 3; ++mode) {
        void *p = mode == 1 ? sig : plain;
        if (mode > 0 && !p) continue;
        hipMemset(p, 0, 16);
        hipDeviceSynchronize();
        hipEventRecord(t0, a);
        unsigned long long k = 0;
        bool ok = true;
        for (int i = 0; i < n && ok; ++i) {
            hipLaunchKernelGGL(spin, dim3(1), dim3(64), 0, a, 10, dummy);
            if (mode == 0) { hipEventRecord(ea, a); hipStreamWaitEvent(b, ea, 0); }
            else { ++k; ok = hipStreamWriteValue64(a, p, k, 0) == hipSuccess && hipStreamWaitValue64(b, p, k, hipStreamWaitValueGte, 0xFFFFFFFFFFFFFFFFull) == hipSuccess; }
            hipLaunchKernelGGL(spin, dim3(1), dim3(64), 0, b, 10, dummy);
            if (mode == 0) { hipEventRecord(eb, b); hipStreamWaitEvent(a, eb, 0); }
            else { ok = ok && hipStreamWriteValue64(b, (char *)p + 8, k, 0) == hipSuccess && hipStreamWaitValue64(a, (char *)p + 8, k, hipStreamWaitValueGte, 0xFFFFFFFFFFFFFFFFull) == hipSuccess; }
        }
        hipEventRecord(t1, a);
        hipError_t e = hipDeviceSynchronize();
        float ms = 0; hipEventElapsedTime(&ms, t0, t1);
        printf("%s: ok=%d sync=%d  %.2f us per round trip (2 hand-offs + 2 trivial kernels)\n", mode == 0 ? "events" : mode == 1 ? "memops on signal memory" : "memops on plain memory", (int)ok, (int)e, ms * 1e3 / n);
        (void)hipGetLastError();
    }
    return 0;
}

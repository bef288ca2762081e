// tools/micro/alloc_kinds.hip -- does HOW a 2.6-GB buffer is allocated decide which of the two speeds (profiles/r03_placement.md) a multi-stream read kernel sees?
// hipMalloc | hipMemCreate + hipMemMap (virtual memory management, 2-MB / 1-GB aligned VA) | hipMallocAsync (pool) | hipExtMallocWithFlags(fine-grained);
// six buffers of each kind, a kernel that reads three streams at once (8 : 2 : 1 bytes, like values / 16-bit ids / 8-bit ids), interleaved timing.
// hipcc --offload-arch=gfx950 -O3 tools/micro/alloc_kinds.hip -o build/micro/alloc_kinds
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(e) do { hipError_t r_ = (e); if (r_ != hipSuccess) { printf("%s: %s\n", #e, hipGetErrorString(r_)); return 1; } } while (0)

// one wave per 16-row "block": 1024 B of values + 256 B of ids + 128 B of narrow ids per step, 20 steps, blocks in address order
__global__ __launch_bounds__(256) void mix(const double *v, const unsigned short *c16, const unsigned char *c8, size_t nblk, double *out)
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
    if (acc == 12345.678) out[b] = acc;
}

struct Buf { char *p; int kind; hipMemGenericAllocationHandle_t h; size_t size; };

static int make(int kind, size_t bytes, Buf &b)
{
    b.kind = kind; b.p = nullptr; b.size = bytes;
    if (kind == 0) { CK(hipMalloc((void **)&b.p, bytes)); }
    else if (kind == 1 || kind == 2) {
        hipMemAllocationProp prop = {};
        prop.type = hipMemAllocationTypePinned; prop.location.type = hipMemLocationTypeDevice; prop.location.id = 0;
        size_t gran = 0;
        CK(hipMemGetAllocationGranularity(&gran, &prop, hipMemAllocationGranularityRecommended));
        const size_t sz = (bytes + gran - 1) / gran * gran;
        b.size = sz;
        CK(hipMemCreate(&b.h, sz, &prop, 0));
        CK(hipMemAddressReserve((void **)&b.p, sz, kind == 2 ? (size_t)1 << 30 : 0, nullptr, 0));
        CK(hipMemMap(b.p, sz, 0, b.h, 0));
        hipMemAccessDesc acc = {};
        acc.location = prop.location; acc.flags = hipMemAccessFlagsProtReadWrite;
        CK(hipMemSetAccess(b.p, sz, &acc, 1));
    } else if (kind == 3) { CK(hipMallocAsync((void **)&b.p, bytes, 0)); CK(hipStreamSynchronize(0)); }
    else { CK(hipExtMallocWithFlags((void **)&b.p, bytes, hipDeviceMallocFinegrained)); }
    CK(hipMemset(b.p, 1, bytes));
    return 0;
}

int main()
{
    const size_t nblk = 75000;                                     // 75 k blocks x 20 steps x (1024 + 256 + 128) B = 2.1 GB
    const size_t vb = nblk * 20 * 1024, cb = nblk * 20 * 256, nb8 = nblk * 20 * 128, total = vb + cb + nb8;
    const char *names[5] = {"hipMalloc", "hipMemCreate/Map", "hipMemCreate/Map, VA 1 GiB aligned", "hipMallocAsync", "hipExtMallocWithFlags(fine-grained)"};
    double *out; CK(hipMalloc((void **)&out, nblk * 8));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    std::vector<Buf> bufs;
    for (int rep = 0; rep < 6; ++rep)
        for (int kind = 0; kind < 5; ++kind) { Buf b; if (make(kind, total, b)) return 1; bufs.push_back(b); }
    for (int rnd = 0; rnd < 2; ++rnd)
        for (int kind = 0; kind < 5; ++kind) {
            printf("round %d %-38s", rnd, names[kind]);
            for (auto &b : bufs) {
                if (b.kind != kind) continue;
                const double *v = (const double *)b.p; const unsigned short *c16 = (const unsigned short *)(b.p + vb); const unsigned char *c8 = (const unsigned char *)(b.p + vb + cb);
                for (int i = 0; i < 3; ++i) hipLaunchKernelGGL(mix, dim3((unsigned)((nblk + 3) / 4)), dim3(256), 0, 0, v, c16, c8, nblk, out);
                CK(hipEventRecord(e0, 0));
                for (int i = 0; i < 100; ++i) hipLaunchKernelGGL(mix, dim3((unsigned)((nblk + 3) / 4)), dim3(256), 0, 0, v, c16, c8, nblk, out);
                CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
                float ms; CK(hipEventElapsedTime(&ms, e0, e1));
                printf("  %.4f ms (%.2f TB/s)", ms / 100, total / (ms / 100) / 1e9);
            }
            printf("\n");
        }
    return 0;
}

// tools/micro/cumask.hip -- can a compute stream be kept off a few CUs so that a kernel with the footprint of RCCL's (256 threads, 280
// registers: one wave per SIMD and nothing beside five 88-register waves) starts at once instead of after the product kernel has drained?
// hipcc --offload-arch=gfx950 -O2 tools/micro/cumask.hip -o /tmp/cumask && /tmp/cumask [reserved CUs per XCD = 2]
//   busy kernel : 256-thread workgroups at ~88 registers, each spinning 20 us, 20 per CU (five rounds at a residency of 5): a stand-in for
//                 a product kernel that fills the device; every workgroup records the CU it ran on
//   fat kernel  : 16 workgroups on a high-priority stream, launched 5 us after the busy kernel; records when it STARTED (100 MHz wall clock)
// printed: CUs the busy kernel used, delay of the fat kernel's first / last workgroup, for an unmasked and a masked compute stream.
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <cstdio>
#include <cstdlib>
#include <set>
#include <vector>
#define CK(e) do { hipError_t r_ = (e); if (r_ != hipSuccess) { printf("%s: %s\n", #e, hipGetErrorString(r_)); return 1; } } while (0)

__device__ unsigned cu_of()
{
    unsigned hw, xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    return ((xcc & 0xf) << 16) | (hw & 0xff00);          // XCC | SE, SH, CU bits of HW_ID
}

__global__ __launch_bounds__(256) void busy(long long ticks, unsigned *where, long long *t_start)
{
    asm volatile("v_mov_b32 v87, 0" ::: "v87");          // 88 registers, as the step kernel
    const long long t0 = wall_clock64();
    if (threadIdx.x == 0) { where[blockIdx.x] = cu_of(); if (blockIdx.x == 0) *t_start = t0; }
    while (wall_clock64() - t0 < ticks) __builtin_amdgcn_s_sleep(8);
}

__global__ __launch_bounds__(256) void fat(long long ticks, unsigned *where, long long *started)
{
    __shared__ int pad[19744 / 4];
    asm volatile("v_mov_b32 v255, 0" ::: "v255");
    asm volatile("v_accvgpr_write_b32 a23, 0" ::: "a23");
    pad[threadIdx.x] = (int)ticks;
    const long long t0 = wall_clock64();
    if (threadIdx.x == 0) { where[blockIdx.x] = cu_of(); started[blockIdx.x] = t0; }
    while (wall_clock64() - t0 < ticks) __builtin_amdgcn_s_sleep(8);
    if (ticks < 0) where[threadIdx.x] = pad[255 - threadIdx.x];
}

__global__ void spin1(long long ticks) { const long long t0 = wall_clock64(); while (wall_clock64() - t0 < ticks) __builtin_amdgcn_s_sleep(8); }

int main(int argc, char **argv)
{
    const int per_xcd = argc > 1 ? atoi(argv[1]) : 2;
    hipDeviceProp_t prop; CK(hipGetDeviceProperties(&prop, 0));
    const int cus = prop.multiProcessorCount, G = cus * 20, F = 16;
    unsigned *where, *fwhere; long long *t_start, *fstart;
    CK(hipMalloc(&where, G * 4)); CK(hipMalloc(&fwhere, 1024)); CK(hipMalloc(&t_start, 8)); CK(hipMalloc(&fstart, F * 8));
    int lo, hi; CK(hipDeviceGetStreamPriorityRange(&lo, &hi));
    hipStream_t comm; CK(hipStreamCreateWithPriority(&comm, hipStreamNonBlocking, hi));
    for (int masked = 0; masked < 2; ++masked) {
        hipStream_t cs;
        if (masked) {
            // KFD deals the mask bits to the XCDs round-robin (bit i -> XCD i % 8), so the top 8 k bits are k CUs of every XCD
            std::vector<uint32_t> m((cus + 31) / 32, 0xFFFFFFFFu);
            for (int b = cus - 8 * per_xcd; b < cus; ++b) m[b / 32] &= ~(1u << (b % 32));
            CK(hipExtStreamCreateWithCUMask(&cs, (uint32_t)m.size(), m.data()));
        } else CK(hipStreamCreateWithFlags(&cs, hipStreamNonBlocking));
        for (int rep = 0; rep < 3; ++rep) {
            CK(hipMemset(fstart, 0, F * 8));
            CK(hipDeviceSynchronize());
            hipLaunchKernelGGL(busy, dim3(G), dim3(256), 0, cs, 2000ll, where, t_start);
            hipLaunchKernelGGL(spin1, dim3(1), dim3(64), 0, comm, 500ll);
            hipLaunchKernelGGL(fat, dim3(F), dim3(256), 0, comm, 1000ll, fwhere, fstart);
            CK(hipDeviceSynchronize());
            std::vector<unsigned> w(G), fw(F); std::vector<long long> fs(F); long long t0;
            CK(hipMemcpy(w.data(), where, G * 4, hipMemcpyDeviceToHost)); CK(hipMemcpy(fw.data(), fwhere, F * 4, hipMemcpyDeviceToHost));
            CK(hipMemcpy(fs.data(), fstart, F * 8, hipMemcpyDeviceToHost)); CK(hipMemcpy(&t0, t_start, 8, hipMemcpyDeviceToHost));
            std::set<unsigned> used(w.begin(), w.end()), fused(fw.begin(), fw.end());
            int per[8] = {0}; for (unsigned u : used) per[(u >> 16) & 7]++;
            long long first = 1ll << 60, last = 0; for (long long s : fs) { first = std::min(first, s - t0); last = std::max(last, s - t0); }
            int shared = 0; for (unsigned u : fused) shared += used.count(u);
            printf("%s rep %d: busy kernel on %zu CUs (per XCD %d %d %d %d %d %d %d %d); fat kernel's workgroups started %.1f .. %.1f us after the busy kernel's first, on %zu CUs of which %d also ran busy workgroups\n",
                   masked ? "masked  " : "unmasked", rep, used.size(), per[0], per[1], per[2], per[3], per[4], per[5], per[6], per[7], first / 100.0, last / 100.0, fused.size(), shared);
        }
        CK(hipStreamDestroy(cs));
    }
    return 0;
}

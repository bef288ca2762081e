// mgx.hip -- the direct exchange of the multi-GPU step (multigpu.cpp "push" mode): every rank STORES its slice of y into every rank's
// gather buffer through peer-mapped pointers (xGMI: one hop, all seven links of a GPU at once) and then stores a sequence number into
// the receiver's `arrived[sender]` word.  No reference counterpart (the reference is single-GPU: src/main_f64.cu:102-168).
//
// Why not only RCCL: its collective kernels on gfx950 (rcclGenericKernel<1|2|4>: 256 threads, 261-280 registers per lane, 19.7 KB of
// LDS -- read from the code object in this image's librccl.so) need a SIMD's register file half empty, and the product kernel keeps five
// 88-register waves on every SIMD and refills each slot the moment it frees: a kernel of that footprint on the highest-priority stream
// starts only when the product has drained (tools/micro/cumask.hip: 19-80 us late; tools/fat_exchange_probe.sh: the step becomes
// product + exchange, 117 us instead of 79).  The kernels below hold 256 threads x <= 32 registers and fit beside the product on any CU.
#include <hip/hip_runtime.h>

#include <string>

#include "plan.hpp"
#include "mgx.hpp"
#include "mgx_device.hpp"

namespace dasp {

#define HIP_TRYX(expr)                                                                              \
    do {                                                                                            \
        hipError_t e_ = (expr);                                                                     \
        if (e_ != hipSuccess) { set_error(std::string(#expr) + ": " + hipGetErrorString(e_)); return DASP_ERR_HIP; } \
    } while (0)

// dasp_mg_push_kernel: grid = wgs workgroups; [optionally wait for `ready` >= ready_need: the products of this step are complete], then
// every workgroup sends its part of the slice (mgx_device.hpp)
__global__ __launch_bounds__(256) void dasp_mg_push_kernel(MgPushArgs a)
{
    if (a.ready_need) {
        if (threadIdx.x == 0) {
            const long long t0 = wall_clock64();
            while (__hip_atomic_load(a.ready, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) < a.ready_need) {
                __builtin_amdgcn_s_sleep(8);
                if (wall_clock64() - t0 > a.timeout) { __hip_atomic_store(a.err, 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM); break; }
            }
        }
        __syncthreads();
    }
    mg_push_part(a, (int)blockIdx.x, a.wgs);
}

// one wave: lane r < world waits for arrived[r] >= seq (the senders' flags), then -- fused step -- `gathered` = step for the waiting
// workgroups of the step kernel (which acquire on their own; in the two-launch form the next kernel on the stream does, at its start).
// A wait that times out sets the sticky error word (3) and goes on: nothing hangs.
__global__ void dasp_mg_arrived_kernel(const unsigned long long *arrived, int world, unsigned long long seq, unsigned long long *gathered,
                                       unsigned long long step, long long timeout, int *err, int skip)
{
    const int lane = threadIdx.x;
    const long long t0 = wall_clock64();
    bool late = false;
    for (int r0 = 0; r0 < world; r0 += 64) {
        const int r = r0 + lane;
        for (;;) {
            const bool ok = r >= world || r == skip || __hip_atomic_load(arrived + r, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) >= seq;
            if (__all(ok)) break;
            __builtin_amdgcn_s_sleep(4);
            if (wall_clock64() - t0 > timeout) { late = true; break; }
        }
    }
    if (lane == 0) {
        if (late) __hip_atomic_store(err, 3, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        if (gathered) __hip_atomic_store(gathered, step, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }
}

int launch_mg_push(const MgPushArgs &a, void *stream)
{
    if (a.n_dst <= 0 || a.wgs <= 0) return DASP_OK;
    hipLaunchKernelGGL(dasp_mg_push_kernel, dim3(a.wgs), dim3(256), 0, static_cast<hipStream_t>(stream), a);
    HIP_TRYX(hipGetLastError());
    return DASP_OK;
}

int launch_mg_arrived(const void *arrived, int world, unsigned long long seq, void *gathered, unsigned long long step, long long timeout_ticks,
                      void *err, void *stream, int skip)
{
    hipLaunchKernelGGL(dasp_mg_arrived_kernel, dim3(1), dim3(64), 0, static_cast<hipStream_t>(stream), static_cast<const unsigned long long *>(arrived), world,
                       seq, static_cast<unsigned long long *>(gathered), step, timeout_ticks, static_cast<int *>(err), skip);
    HIP_TRYX(hipGetLastError());
    return DASP_OK;
}

}  // namespace dasp

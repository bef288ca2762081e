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

namespace dasp {

#define HIP_TRYX(expr)                                                                              \
    do {                                                                                            \
        hipError_t e_ = (expr);                                                                     \
        if (e_ != hipSuccess) { set_error(std::string(#expr) + ": " + hipGetErrorString(e_)); return DASP_ERR_HIP; } \
    } while (0)

// grid = wgs workgroups.  Workgroup w: [optionally wait for `ready` >= ready_need: the products of this step are complete] load part w
// of the slice ONCE and store it to every destination, then count itself at every destination's counter; the last arrival at a
// destination's counter publishes `seq` in that destination's flag word.  16-byte loads / stores, up to 4 loads and 4 x n_dst stores in
// flight per lane: with 256 workgroups a 2-MB slice is ONE round (first layout: a workgroup per (destination, part), the slice read
// n_dst times in 4-8 dependent rounds -- 25 us under the running product instead of 9 alone).
// No fences: a release fence is a write-back of the whole L2 (buffer_wbl2) and an acquire an invalidate, per wave, under the running
// product (first version, 16 workgroups per destination: the step 108 us instead of 75; 32: 138 us).  Instead every access is
// system-coherent by itself -- sc0 sc1 loads (the slice was written through by the product's sc0 sc1 stores) and sc0 sc1 write-through
// stores, complete when s_waitcnt vmcnt(0) returns -- and the counters and the flags are relaxed atomics issued after that.
// (as inline assembly: the compiler puts an s_waitcnt vmcnt(0) behind EVERY volatile access, one access in flight per wave)
typedef unsigned v4u __attribute__((ext_vector_type(4)));
__device__ __forceinline__ v4u ld_sys(const v4u *p)
{
    v4u v;
    asm volatile("global_load_dwordx4 %0, %1, off sc0 sc1" : "=v"(v) : "v"(p) : "memory");
    return v;
}
__device__ __forceinline__ void st_sys(v4u *p, v4u v) { asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1" : : "v"(p), "v"(v) : "memory"); }

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
    const size_t n16 = a.bytes >> 4;                                   // the slice is a multiple of 64 elements: of 16 bytes
    const size_t per = (n16 + a.wgs - 1) / a.wgs, i0 = per * blockIdx.x, i1 = i0 + per < n16 ? i0 + per : n16;
    const v4u *src = reinterpret_cast<const v4u *>(a.src);
    // the destination table through the CONSTANT address space: scalar loads (lgkmcnt) -- a vector load of it would make the compiler wait
    // for vmcnt(0), i.e. for the previous destination's stores to be acknowledged, before every destination
    typedef const __attribute__((address_space(4))) MgPushDst *DstTab;
    const DstTab dst = (DstTab)(uintptr_t)a.dst;
    size_t i = i0 + threadIdx.x;
    for (; i + 768 < i1; i += 1024) {                                  // the stores of one round are in flight under the loads of the next
        const v4u v0 = ld_sys(src + i), v1 = ld_sys(src + i + 256), v2 = ld_sys(src + i + 512), v3 = ld_sys(src + i + 768);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        for (int d = 0; d < a.n_dst; ++d) {
            v4u *out = reinterpret_cast<v4u *>(dst[d].data);
            st_sys(out + i, v0); st_sys(out + i + 256, v1); st_sys(out + i + 512, v2); st_sys(out + i + 768, v3);
        }
    }
    for (; i < i1; i += 256) {
        const v4u v = ld_sys(src + i);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        for (int d = 0; d < a.n_dst; ++d) st_sys(reinterpret_cast<v4u *>(dst[d].data) + i, v);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                   // this wave's stores have been performed at their destinations
    __syncthreads();
    for (int d = threadIdx.x; d < a.n_dst; d += 256) {
        const unsigned old = __hip_atomic_fetch_add(a.count + d, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if ((old + 1) % (unsigned)a.wgs == 0)                          // every part for this destination is out
            __hip_atomic_store(a.dst[d].flag, a.seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }
}

// one wave: lane r < world waits for arrived[r] >= seq (the senders' flags), then -- fused step -- `gathered` = step for the waiting
// workgroups of the step kernel (which acquire on their own; in the two-launch form the next kernel on the stream does, at its start).
// A wait that times out sets the sticky error word (3) and goes on: nothing hangs.
__global__ void dasp_mg_arrived_kernel(const unsigned long long *arrived, int world, unsigned long long seq, unsigned long long *gathered,
                                       unsigned long long step, long long timeout, int *err)
{
    const int lane = threadIdx.x;
    const long long t0 = wall_clock64();
    bool late = false;
    for (int r0 = 0; r0 < world; r0 += 64) {
        const int r = r0 + lane;
        for (;;) {
            const bool ok = r >= world || __hip_atomic_load(arrived + r, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) >= seq;
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
                      void *err, void *stream)
{
    hipLaunchKernelGGL(dasp_mg_arrived_kernel, dim3(1), dim3(64), 0, static_cast<hipStream_t>(stream), static_cast<const unsigned long long *>(arrived), world,
                       seq, static_cast<unsigned long long *>(gathered), step, timeout_ticks, static_cast<int *>(err));
    HIP_TRYX(hipGetLastError());
    return DASP_OK;
}

}  // namespace dasp

// mgx_device.hpp -- device side of the direct ("push") exchange: one workgroup's share of sending this rank's slice to every destination.
// Used by dasp_mg_push_kernel (mgx.hip: the exchange as a kernel of its own) and by the head workgroups of the one-stream step kernel
// (mgstep.hip: dasp_mg_step2_kernel).  No reference counterpart (the reference is single-GPU: src/main_f64.cu:102-168).
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>

#include "mgx.hpp"

namespace dasp {

// `parts` workgroups.  Workgroup w: load part w
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

// workgroup `part` of `parts` (256 threads): its share of the slice to every destination, then the destination counters and flags
__device__ __forceinline__ void mg_push_part(const MgPushArgs &a, int part, int parts)
{
    const size_t n16 = a.bytes >> 4;                                   // the slice is a multiple of 64 elements: of 16 bytes
    const size_t per = (n16 + parts - 1) / parts, i0 = per * part, i1 = i0 + per < n16 ? i0 + per : n16;
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
        if ((old + 1) % (unsigned)parts == 0) {                        // every part for this destination is out
            if (a.delay_ticks > 0) {                                   // loopback timing probes only: the links' share of the exchange (100 MHz ticks)
                const long long t0 = wall_clock64();
                while (wall_clock64() - t0 < a.delay_ticks) __builtin_amdgcn_s_sleep(8);
            }
            __hip_atomic_store(a.dst[d].flag, a.seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        }
    }
}

}  // namespace dasp

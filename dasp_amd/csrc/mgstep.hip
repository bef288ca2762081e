// mgstep.hip -- the multi-GPU step as ONE launch (dasp_mg_spmv, f64; driven by multigpu.cpp): kernels, launchers and the host
// tables they need.  No reference counterpart: the reference drives one device (src/main_f64.cu:102-168).
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdlib>
#include <string>
#include <vector>

#include "spmv_device.hpp"
#include "mgx_device.hpp"

namespace dasp {

// ------------------------------------------------------------------ the fused multi-GPU step (dasp_mg_spmv, f64)
// One launch = the whole product of one rank's step.  Grid order: the workgroups of the own-column plan `a` (y = own, written
// through), then a bounded set of PERSISTENT workgroups that wait -- in the kernel -- for two flags and then stride over the
// workgroups of the other-column plan `b` (y += other: exactly the arithmetic of the two-launch form):
//   own_go   : every own-column workgroup that handles a row with other-column nonzeros ("marked", a host-built table) has stored
//              its y.  The own plan's medium blocks are dispatched through a host-built order table -- the blocks holding such rows
//              first, then the rest longest-first as ever (a plain reversal loses 20 us to the long blocks at the tail) -- so the
//              marked workgroups run FIRST and the flag is up after a fraction of the own-column product;
//   gathered : the exchange of the previous step has delivered the other ranks' x.
// So the other-column product overlaps the rest of the own-column product instead of following it, neither a kernel boundary nor a
// stream wait sits between the two, and the last workgroup to finish publishes "y ready" itself.  The waiting workgroups are at most
// max_pollers (multigpu.cpp: one slot per CU is always left free), so they can never fill the device and keep the exchange's kernel
// out; a wait longer than the time-out sets *err and skips the other-column product instead of hanging -- the host then falls back to
// the two-launch form (dasp_mg_check).
struct StepCtl {
    const unsigned long long *gathered;   // device word: step number of the last completed exchange into the gather buffer
    unsigned long long need;              // the other-column product may read the gather buffer once *gathered >= need (0: at once)
    const unsigned char *mark;            // [grid_a] 1: that own-column workgroup stores a row the other-column plan adds to
    const unsigned *mark_members;         // [64] marked workgroups per shard (wg & 63)
    int n_marked, n_mark_shards;          // marked workgroups / non-empty shards among them
    unsigned *mark_shards, *mark_top;     // arrival counters of the marked own-column workgroups (64 shards on lines of their own + top)
    unsigned long long *own_go;           // set to `step` by the last marked workgroup: the other-column product may add into y
    unsigned *all_shards, *all_top;       // arrival counters of ALL workgroups of the launch
    unsigned long long *ready;            // word the exchange waits on: set to `step` when every workgroup of this launch is done
    unsigned long long step;
    int *err;                             // sticky: 1 = a wait timed out
    int grid_a, grid_b, n_poll;           // workgroups of plan a / virtual workgroups of plan b / persistent workgroups serving them
    int poll_at;                          // grid position of the first waiting workgroup (<= grid_a)
    unsigned long long *xcd_fenced;       // [8 x 32] per XCD: the step whose system-scope acquire is done; + 16: claimed
    const int *blk_order;                 // [medium blocks of plan a] dispatch order: the blocks of marked workgroups first
    int sleep;                            // s_sleep(8) repetitions between two polls (~0.2 us each)
    long long timeout;                    // 100 MHz ticks after which a waiting workgroup gives up
    int count_all;                        // 1: every workgroup arrives at the `ready` counter; 0: the caller publishes "y ready" behind the launch (stream order)
};

// arrival of a workgroup at a two-level counter: true for the one that arrives last.  64 sharded counters, each on a 128-byte line
// of its own, then one top counter -- same-address atomics serialise at the memory side (one flat counter: ~50 ns per arrival,
// 180 us for the 3800 workgroups of an 8-way HV15R slice).  `members` = arrivals expected at this shard, `nshards` = non-empty
// shards.  Every counter returns to 0 with its last arrival.
constexpr int kArriveShards = 64, kShardStride = 32;        // in 4-byte words
__device__ __forceinline__ bool arrive_last(unsigned *shards, unsigned *top, int sh, unsigned members, unsigned nshards)
{
    if (__hip_atomic_fetch_add(shards + sh * kShardStride, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) + 1 != members) return false;
    __hip_atomic_store(shards + sh * kShardStride, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (__hip_atomic_fetch_add(top, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) + 1 != nshards) return false;
    __hip_atomic_store(top, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    return true;
}
// members of shard `sh` among `total` consecutively numbered arrivals
__device__ __forceinline__ unsigned shard_members(int total, int sh) { return (unsigned)((total - sh + kArriveShards - 1) / kArriveShards); }

template <bool NT>
__global__ __launch_bounds__(256, 6) void dasp_mg_step_kernel(DevArgs a, DevArgs b, StepCtl c)
{
    __shared__ int go;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    // the waiting workgroups stand at [poll_at, poll_at + n_poll) of the grid (poll_at = grid_a: last): early enough for the other-column
    // product to run in the shadow of the own-column one instead of behind it
    const int gi = blockIdx.x;
    const bool own_wg = gi < c.poll_at || gi >= c.poll_at + c.n_poll;
    const int wg = gi < c.poll_at ? gi : own_wg ? gi - c.n_poll : c.grid_a + (gi - c.poll_at);      // own-column workgroup number, or grid_a + waiting workgroup number
    const int total = c.grid_a + c.n_poll;
    if (own_wg) {
        plain_wg<double, NT, true, true, 1>(a, wg, wave, lane, c.blk_order);
        // ("ready" published behind the launch: a workgroup nobody waits for just ends -- no wait for its stores, no barrier, no atomic)
        if (!c.count_all && !(c.n_marked > 0 && tab<true>(c.mark, wg))) return;
        // done: every wave's write-through stores acknowledged, then ONE lane counts the workgroup.  Relaxed atomics: the y values went
        // out through sc0 sc1 stores, so an arrival needs no cache write-back or invalidate of its own (an acq_rel add costs every
        // workgroup a buffer_wbl2 + buffer_inv: measured 300 instead of 70 us per step)
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (threadIdx.x == 0) {
            if (c.n_marked > 0 && tab<true>(c.mark, wg) &&
                arrive_last(c.mark_shards, c.mark_top, wg & (kArriveShards - 1), tab<true>(c.mark_members, wg & (kArriveShards - 1)), (unsigned)c.n_mark_shards))
                __hip_atomic_store(c.own_go, c.step, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (c.count_all && arrive_last(c.all_shards, c.all_top, wg & (kArriveShards - 1), shard_members(total, wg & (kArriveShards - 1)),
                                           (unsigned)(total < kArriveShards ? total : kArriveShards)))
                __hip_atomic_store(c.ready, c.step, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
        }
        return;
    }
    if (threadIdx.x == 0) {
        int ok = 1;
        const long long t0 = wall_clock64();
        // relaxed polls (an acquire load would invalidate caches on every iteration, under the running product), ONE acquire at the end
        while ((c.n_marked > 0 && __hip_atomic_load(c.own_go, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < c.step) ||
               (c.need && __hip_atomic_load(c.gathered, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) < c.need)) {
            for (int z = 0; z < c.sleep; ++z) __builtin_amdgcn_s_sleep(8);
            if (wall_clock64() - t0 > c.timeout) { ok = 0; __hip_atomic_store(c.err, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM); break; }
        }
        // the gather buffer was written by another kernel (this device's or, over xGMI, a peer's) while this one ran: acquire at system
        // scope -- ONE workgroup per XCD does it (a system-scope acquire drops what the XCD's L2 holds, under the running own-column product;
        // r3 had every waiting workgroup do it), the others wait for that XCD's word.  No L1 holds a line of the gather buffer yet: only
        // the other-column plan reads it, i.e. workgroups that have passed this point.
        if (ok) {
            unsigned xcc;
            asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
            unsigned long long *w = c.xcd_fenced + 32 * (xcc & 7);
            if (__hip_atomic_load(w, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < c.step) {
                if (__hip_atomic_fetch_max(w + 16, c.step, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < c.step) {
                    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "");
                    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                    __hip_atomic_store(w, c.step, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                } else {
                    const long long t1 = wall_clock64();
                    while (__hip_atomic_load(w, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < c.step && wall_clock64() - t1 < c.timeout) __builtin_amdgcn_s_sleep(2);
                }
            }
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        go = ok;
    }
    __syncthreads();
    if (go)
        for (int v = wg - c.grid_a; v < c.grid_b; v += c.n_poll) plain_wg<double, NT, true, true, 2>(b, v, wave, lane, nullptr);
    if (!c.count_all) return;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (threadIdx.x == 0 && arrive_last(c.all_shards, c.all_top, wg & (kArriveShards - 1), shard_members(total, wg & (kArriveShards - 1)),
                                        (unsigned)(total < kArriveShards ? total : kArriveShards)))
        __hip_atomic_store(c.ready, c.step, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}

// the exchange side of the fused step, on the communication stream: hold the stream until *p >= need (the product's "y ready"),
// and publish a step number behind the exchange.  Plain kernels on plain device words: no stream memory operations (Beta API).
// Neither needs a fence of its own: what the wait kernel orders is the NEXT kernel on its stream (the exchange), which acquires at its
// start like every kernel; what the flag kernel publishes was written by the PREVIOUS kernel on its stream, released at that kernel's
// end -- a release fence in a one-lane kernel is a buffer_wbl2 under the running product (rocprofv3: 6.5 us per flag kernel with it).
__global__ void dasp_mg_wait_kernel(const unsigned long long *p, unsigned long long need, long long timeout, int *err)
{
    if (threadIdx.x != 0) return;
    const long long t0 = wall_clock64();
    while (__hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) < need) {
        __builtin_amdgcn_s_sleep(8);
        if (wall_clock64() - t0 > timeout) { __hip_atomic_store(err, 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM); break; }
    }
}
__global__ void dasp_mg_flag_kernel(unsigned long long *p, unsigned long long v)
{
    if (threadIdx.x == 0) __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}

// ------------------------------------------------------------------ the step on ONE stream (r4: dasp_mg_step2_kernel)
// The rank's slice is ONE plan whose column ids index the gather buffer (every rank's slice of x in equal padded slots).  One launch per
// iteration and nothing else, on the caller's stream only:
//   head workgroups       send the slice the PREVIOUS launch produced (this rank's own slot of the half of the gather buffer that holds
//                         x) into every peer's gather buffer -- direct stores through peer mappings, then the peers' arrival flags
//                         (mg_push_part).  The exchange of step k therefore runs under the product of step k + 1 without a
//                         communication stream, a kernel of its own, or any word the product would have to publish;
//   free workgroups       every workgroup of the plan whose rows read this rank's own columns only, in the plan's order;
//   persistent workgroups (a bounded number, last in the grid) wait for the arrival flags of ALL peers and then stride over the
//                         workgroups whose rows read other ranks' columns ("boundary rows": the host puts those blocks LAST in the
//                         dispatch order, so that their turn comes when the peers' slices have long arrived).
// y goes into this rank's slot of the OTHER half of the gather buffer: it is the next launch's x as it stands.  Every row is computed
// once, by one workgroup, from one plan: no second product, no y +=, no counters, no write-through stores (compare dasp_mg_step_kernel:
// its other-column product was a serial tail of ~10 us behind the own-column product, profiles/r04_multi_gpu_step.md).
// Because every launch waits for all peers' slices of the previous step, no rank can run two steps ahead of another: a peer's stores of
// step k + 1 go to the half this rank reads in step k + 2, never to the half it is reading.
struct Step2Ctl {
    const int *wg_list;                     // [n_free + n_marked] virtual workgroups of the plan: free ones first, boundary ones last
    const int *blk_order;                   // [medium blocks] dispatch order: blocks with boundary rows last
    int n_push, n_free, n_marked, n_total, n_poll;     // list: n_free unmarked, n_marked marked, the other unmarked ones; n_poll 0: marked workgroups wait in place
    const unsigned long long *arrived;      // [world] this rank's arrival flags (fine-grained memory, written by the peers)
    int world, rank;
    unsigned long long need;                // the flags must reach this value (0: x was put in place by the host, nothing to wait for)
    int *err; long long timeout; int sleep;
    int fence_mode;                         // 0: every waiting workgroup acquires at system scope; 1: one per XCD; 2: none (measurement only)
    unsigned long long *xcd_fenced;         // [8 x 32] per XCD: the step whose fence is done; + 16: the step whose fence is claimed
    unsigned long long step;
};

// lane 0 of the calling wave ends up knowing whether every peer's slice of the previous step has arrived (true) or the wait timed out
__device__ __forceinline__ bool step2_wait(const Step2Ctl &c, int lane)
{
    bool late = false;
    if (c.need) {
        const long long t0 = wall_clock64();
        for (int r0 = 0; r0 < c.world && !late; r0 += 64) {
            const int r = r0 + lane;
            for (;;) {       // relaxed polls
                const bool ok = r >= c.world || r == c.rank || __hip_atomic_load(c.arrived + r, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) >= c.need;
                if (__all(ok)) break;
                for (int z = 0; z < c.sleep; ++z) __builtin_amdgcn_s_sleep(8);
                if (wall_clock64() - t0 > c.timeout) { late = true; break; }
            }
        }
    }
    // The peers' slices were stored by other devices while this kernel ran: acquire at system scope.  A system-scope acquire throws away
    // what the XCD's L2 holds (buffer_inv sc0 sc1) -- the x this rank's own rows are gathering from.  One per XCD is all the memory model
    // asks for of the L2s: an L2 belongs to an XCD.  fence_mode 1: the first waiting workgroup of each XCD to get here claims
    // the fence, performs it and raises that XCD's word to the step number; the others wait for the word (a few hundred cycles).
    if (c.fence_mode == 0) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "");
    else if ((c.fence_mode == 1 || c.fence_mode == 3) && c.need && !late) {      // (3: measurement -- the per-XCD fence without the L1 invalidates)
        unsigned xcc;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
        unsigned long long *w = c.xcd_fenced + 32 * (xcc & 7);            // two 128-byte lines per XCD: fenced, claimed
        if (lane == 0 && __hip_atomic_load(w, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < c.step) {
            if (__hip_atomic_fetch_max(w + 16, c.step, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < c.step) {      // this workgroup fences for its XCD
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "");
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                __hip_atomic_store(w, c.step, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            } else {
                const long long t1 = wall_clock64();
                while (__hip_atomic_load(w, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < c.step) {
                    __builtin_amdgcn_s_sleep(2);
                    if (wall_clock64() - t1 > c.timeout) break;          // (the fencing workgroup cannot get lost; a bound all the same)
                }
            }
        }
    }
    // ... and every waiting workgroup drops its own CU's L1 (an agent-scope acquire: buffer_inv sc1, the L2 stays): a pad's clamped gather
    // reads x[0] -- rank 0's slot -- whatever the row, so a line of a peer's slot CAN sit in an L1 from before the flags went up
    if ((c.fence_mode == 1 || c.fence_mode == 0) && c.need) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (lane == 0 && late) __hip_atomic_store(c.err, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    return !late;
}

template <bool NT>
__global__ __launch_bounds__(256, 6) void dasp_mg_step2_kernel(DevArgs a, MgPushArgs push, Step2Ctl c)
{
    __shared__ int go;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    int wg = blockIdx.x;
    if (wg < c.n_push) { mg_push_part(push, wg, c.n_push); return; }
    wg -= c.n_push;
    if (c.n_poll == 0) {
        // every workgroup of the list where it stands; the marked ones [n_free, n_free + n_marked) wait for the peers' slices by themselves
        if (wg >= c.n_free && wg < c.n_free + c.n_marked) {
            if (wave == 0) { const bool ok = step2_wait(c, lane); if (lane == 0) go = ok ? 1 : 0; }
            __syncthreads();
            if (!go) return;
        }
        plain_wg<double, NT, true, true, 3>(a, tab<true>(c.wg_list, wg), wave, lane, c.blk_order);
        return;
    }
    // several ranks on one device: the marked workgroups through a bounded set of persistent workgroups at the END of the grid
    const int n_unmarked = c.n_total - c.n_marked;
    if (wg < n_unmarked) { plain_wg<double, NT, true, true, 3>(a, tab<true>(c.wg_list, wg < c.n_free ? wg : wg + c.n_marked), wave, lane, c.blk_order); return; }
    wg -= n_unmarked;
    if (wave == 0) { const bool ok = step2_wait(c, lane); if (lane == 0) go = ok ? 1 : 0; }
    __syncthreads();
    if (go)
        for (int v = wg; v < c.n_marked; v += c.n_poll) plain_wg<double, NT, true, true, 3>(a, tab<true>(c.wg_list, c.n_free + v), wave, lane, c.blk_order);
}

#define HIP_TRY(expr)                                                                          \
    do {                                                                                       \
        hipError_t e_ = (expr);                                                                \
        if (e_ != hipSuccess) {                                                                \
            set_error(std::string(#expr) + ": " + hipGetErrorString(e_));                      \
            return e_ == hipErrorNoDevice ? DASP_ERR_NO_DEVICE : DASP_ERR_HIP;                 \
        }                                                                                      \
    } while (0)

int launch_mg_step(Plan &own, Plan *other, const void *x_own, const void *x_gathered, void *y, const MgStepCtl &h, void *stream)
{
    if (!mg_step_supported(own, other)) { set_error("plans do not qualify for the fused multi-GPU step"); return DASP_ERR_STATE; }
    DevArgs a = own.dev->args, b = other ? other->dev->args : own.dev->args;
    a.x = x_own; a.y = y; a.acc = 0;
    b.x = x_gathered; b.y = y; b.acc = 0;
    // the step kernel's own geometry: one medium block per wave in table order (mg_step_marks assumes it), whatever upload_plan chose
    a.wg_med = (a.n_blocks + kWavesPerWG - 1) / kWavesPerWG; a.xcd_on = 0;
    b.wg_med = (b.n_blocks + kWavesPerWG - 1) / kWavesPerWG; b.xcd_on = 0;
    StepCtl c{};
    char *w = static_cast<char *>(h.words);
    c.mark_shards = reinterpret_cast<unsigned *>(w); c.all_shards = reinterpret_cast<unsigned *>(w + 8192);
    c.mark_top = reinterpret_cast<unsigned *>(w + 16384); c.all_top = reinterpret_cast<unsigned *>(w + 16384 + 256);
    c.gathered = reinterpret_cast<const unsigned long long *>(w + kMgWordGathered); c.need = other ? h.need : 0;
    c.own_go = reinterpret_cast<unsigned long long *>(w + kMgWordOwnGo);
    c.ready = reinterpret_cast<unsigned long long *>(w + kMgWordReady); c.step = h.step; c.err = h.err ? static_cast<int *>(h.err) : reinterpret_cast<int *>(w + kMgWordErr);
    c.grid_a = a.wg_long + a.wg_med + a.wg_short;
    c.grid_b = other ? b.wg_long + b.wg_med + b.wg_short : 0;
    c.n_poll = std::min(c.grid_b, std::max(1, h.max_pollers));
    c.poll_at = std::max(0, std::min(c.grid_a, (int)((double)c.grid_a * h.poll_at)));
    c.xcd_fenced = reinterpret_cast<unsigned long long *>(w + kMgWordXcd);
    c.mark = static_cast<const unsigned char *>(h.mark); c.mark_members = static_cast<const unsigned *>(h.mark_members);
    c.n_marked = other ? h.n_marked : 0; c.n_mark_shards = h.n_mark_shards;
    c.blk_order = static_cast<const int *>(h.blk_order);
    c.sleep = std::max(1, h.poll_sleep); c.timeout = h.timeout_ticks;
    c.count_all = h.ready_by_event ? 0 : 1;
    const int grid = c.grid_a + c.n_poll;
    if (grid <= 0) { set_error("empty step"); return DASP_ERR_STATE; }
    hipStream_t s = static_cast<hipStream_t>(stream);
    // the own-column plan decides the cache policy of the streamed tiles (the other-column plan is a few per cent of the bytes)
    if (own.dev->nt) hipLaunchKernelGGL((dasp_mg_step_kernel<true>), dim3(grid), dim3(256), 0, s, a, b, c);
    else hipLaunchKernelGGL((dasp_mg_step_kernel<false>), dim3(grid), dim3(256), 0, s, a, b, c);
    HIP_TRY(hipGetLastError());
    return DASP_OK;
}

// workgroups of the step kernel one CU holds at a time (multigpu.cpp keeps one of them free of waiting workgroups)
int mg_step_resident_per_cu()
{
    int a = 0, b = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&a, reinterpret_cast<const void *>(&dasp_mg_step_kernel<true>), 256, 0) != hipSuccess) a = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&b, reinterpret_cast<const void *>(&dasp_mg_step_kernel<false>), 256, 0) != hipSuccess) b = 0;
    int c2 = 0, d2 = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&c2, reinterpret_cast<const void *>(&dasp_mg_step2_kernel<true>), 256, 0) != hipSuccess) c2 = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&d2, reinterpret_cast<const void *>(&dasp_mg_step2_kernel<false>), 256, 0) != hipSuccess) d2 = 0;
    (void)hipGetLastError();
    return std::min(std::min(a, b), std::min(c2, d2));
}

int launch_mg_wait(const void *word, unsigned long long need, long long timeout_ticks, void *err, void *stream)
{
    hipLaunchKernelGGL(dasp_mg_wait_kernel, dim3(1), dim3(64), 0, static_cast<hipStream_t>(stream), static_cast<const unsigned long long *>(word), need,
                       timeout_ticks, static_cast<int *>(err));
    HIP_TRY(hipGetLastError());
    return DASP_OK;
}

int launch_mg_flag(void *word, unsigned long long value, void *stream)
{
    hipLaunchKernelGGL(dasp_mg_flag_kernel, dim3(1), dim3(64), 0, static_cast<hipStream_t>(stream), static_cast<unsigned long long *>(word), value);
    HIP_TRY(hipGetLastError());
    return DASP_OK;
}

int launch_mg_step2(Plan &plan, const void *x, void *y, const MgStep2Ctl &h, const MgPushArgs &push, void *stream)
{
    if (!mg_step_supported(plan, nullptr)) { set_error("plan does not qualify for the one-stream multi-GPU step"); return DASP_ERR_STATE; }
    DevArgs a = plan.dev->args;
    a.x = x; a.y = y; a.acc = 0;
    a.wg_med = (a.n_blocks + kWavesPerWG - 1) / kWavesPerWG; a.xcd_on = 0;      // one medium block per wave through the order table (mg_step_marks assumes it)
    Step2Ctl c{};
    c.wg_list = static_cast<const int *>(h.wg_list); c.blk_order = static_cast<const int *>(h.blk_order);
    c.n_push = h.n_push; c.n_free = h.n_free; c.n_marked = h.n_marked; c.n_total = h.n_total;
    // bounded mode: at least one persistent workgroup -- it is also what holds the launch until every peer's slice is in
    c.n_poll = h.max_pollers > 0 ? std::max(1, std::min(h.n_marked, h.max_pollers)) : 0;
    c.arrived = static_cast<const unsigned long long *>(h.arrived); c.world = h.world; c.rank = h.rank; c.need = h.need;
    c.err = static_cast<int *>(h.err); c.timeout = h.timeout_ticks; c.sleep = std::max(1, h.poll_sleep);
    c.fence_mode = h.fence_mode; c.xcd_fenced = static_cast<unsigned long long *>(h.xcd_fenced); c.step = h.step;
    if (c.n_total != a.wg_long + a.wg_med + a.wg_short || c.n_free + c.n_marked > c.n_total) { set_error("one-stream step: the workgroup list does not match the plan's grid"); return DASP_ERR_STATE; }
    // in-place mode with no marked workgroup at all (a rank without boundary rows): ONE extra workgroup still waits for every peer, so that
    // no rank can run two steps ahead of another
    if (c.n_poll == 0 && c.n_marked == 0 && c.need) c.n_poll = 1;
    const int grid = c.n_push + (c.n_poll ? c.n_total - c.n_marked + c.n_poll : c.n_total);
    hipStream_t s = static_cast<hipStream_t>(stream);
    if (plan.dev->nt) hipLaunchKernelGGL((dasp_mg_step2_kernel<true>), dim3(grid), dim3(256), 0, s, a, push, c);
    else hipLaunchKernelGGL((dasp_mg_step2_kernel<false>), dim3(grid), dim3(256), 0, s, a, push, c);
    HIP_TRY(hipGetLastError());
    return DASP_OK;
}

}  // namespace dasp

// mgstep.hip -- the multi-GPU step as ONE launch (dasp_mg_spmv, f64; driven by multigpu.cpp): kernels, launchers and the host
// tables they need.  No reference counterpart: the reference drives one device (src/main_f64.cu:102-168).
#include <hip/hip_runtime.h>

#include <algorithm>
#include <string>
#include <vector>

#include "spmv_device.hpp"

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
    const int *blk_order;                 // [medium blocks of plan a] dispatch order: the blocks of marked workgroups first
    int sleep;                            // s_sleep(8) repetitions between two polls (~0.2 us each)
    long long timeout;                    // 100 MHz ticks after which a waiting workgroup gives up
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
__global__ __launch_bounds__(256, kMinWavesPlain) void dasp_mg_step_kernel(DevArgs a, DevArgs b, StepCtl c)
{
    __shared__ int go;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int wg = blockIdx.x;
    const int total = c.grid_a + c.n_poll;
    if (wg < c.grid_a) {
        plain_wg<double, NT, true, true, 1>(a, wg, wave, lane, c.blk_order);
        // done: every wave's write-through stores acknowledged, then ONE lane counts the workgroup.  Relaxed atomics: the y values went
        // out through sc0 sc1 stores, so an arrival needs no cache write-back or invalidate of its own (an acq_rel add costs every
        // workgroup a buffer_wbl2 + buffer_inv: measured 300 instead of 70 us per step)
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (threadIdx.x == 0) {
            if (c.n_marked > 0 && tab<true>(c.mark, wg) &&
                arrive_last(c.mark_shards, c.mark_top, wg & (kArriveShards - 1), tab<true>(c.mark_members, wg & (kArriveShards - 1)), (unsigned)c.n_mark_shards))
                __hip_atomic_store(c.own_go, c.step, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (arrive_last(c.all_shards, c.all_top, wg & (kArriveShards - 1), shard_members(total, wg & (kArriveShards - 1)),
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
        // the gather buffer was written by another kernel (this device's or, over xGMI, a peer's) while this one ran
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        go = ok;
    }
    __syncthreads();
    if (go)
        for (int v = wg - c.grid_a; v < c.grid_b; v += c.n_poll) plain_wg<double, NT, true, true, 2>(b, v, wave, lane, nullptr);
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

#define HIP_TRY(expr)                                                                          \
    do {                                                                                       \
        hipError_t e_ = (expr);                                                                \
        if (e_ != hipSuccess) {                                                                \
            set_error(std::string(#expr) + ": " + hipGetErrorString(e_));                      \
            return e_ == hipErrorNoDevice ? DASP_ERR_NO_DEVICE : DASP_ERR_HIP;                 \
        }                                                                                      \
    } while (0)

// ---- fused multi-GPU step (multigpu.cpp).  Which plans qualify: f64, uploaded, no x windows, no column panels, 16-bit ids (the one
// instantiation of the step kernel), no long row cut into several pieces (their stage 2 would run behind the launch that publishes
// "y ready").  `other` may be null (no nonzero outside the rank's own columns).
bool mg_step_supported(const Plan &own, const Plan *other)
{
    auto ok = [](const Plan &p) {
        // 16-bit ids -- or no MFMA block at all (every medium row stored as a slab: nothing reads the id planes)
        return p.precision == 64 && p.dev && p.dev->arena && p.panels.empty() && !p.windowed && (p.cid16 || p.stats.n_med_blocks == 0) && p.dev->args.n_multi == 0;
    };
    return ok(own) && (!other || ok(*other));
}

void mg_step_marks(const Plan &p, const unsigned char *has_other, std::vector<unsigned char> &mark, std::vector<int> &blk_order)
{
    // the launch grid of upload_plan / dasp_mg_step_kernel: [ long pieces | medium blocks through blk_order | short tiles ], 4 units per workgroup
    const int n_pieces = (int)p.piece_dst.size(), n_blocks = p.stats.n_med_blocks, n_tiles = p.stats.n_short_tiles;
    const int wg_long = (n_pieces + kWavesPerWG - 1) / kWavesPerWG, wg_med = (n_blocks + kWavesPerWG - 1) / kWavesPerWG,
              wg_short = (n_tiles + kWavesPerWG - 1) / kWavesPerWG;
    mark.assign((size_t)wg_long + wg_med + wg_short, 0);
    for (int q = 0; q < n_pieces; ++q) {
        const int dst = p.piece_dst[(size_t)q];
        if (dst >= 0 && has_other[dst]) mark[(size_t)q / kWavesPerWG] = 1;
    }
    // medium blocks holding a row the other-column plan adds to go first, everything else keeps the stored (longest-first) order
    std::vector<unsigned char> hot((size_t)n_blocks, 0);
    for (int b = 0; b < n_blocks; ++b)
        for (int i = 0; i < kMedRows && b * kMedRows + i < p.n_mfma_rows; ++i)
            if (has_other[p.order[(size_t)p.med_slot0 + (size_t)b * kMedRows + i]]) { hot[(size_t)b] = 1; break; }
    blk_order.clear(); blk_order.reserve((size_t)n_blocks);
    for (int b = 0; b < n_blocks; ++b) if (hot[(size_t)b]) blk_order.push_back(b);
    const int n_hot = (int)blk_order.size();
    for (int b = 0; b < n_blocks; ++b) if (!hot[(size_t)b]) blk_order.push_back(b);
    for (int q = 0; q < n_hot; ++q) mark[(size_t)wg_long + q / kWavesPerWG] = 1;
    const int SR = p.geo.short_rows;
    for (int g = 0; g < kNumShortGroups; ++g) {
        const ShortGroup &G = p.grp[g];
        for (int lt = 0; lt < G.tiles; ++lt) {
            const int t = G.tile0 + lt;
            for (int tt = lt * SR; tt < std::min(G.count, (lt + 1) * SR); ++tt) {
                const int slot = g < 5 ? G.map.slot(tt) : G.map.base[0] + tt;      // short_rows / slab_rows
                if (has_other[p.order[(size_t)slot]]) { mark[(size_t)wg_long + wg_med + t / kWavesPerWG] = 1; break; }
            }
        }
    }
}

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
#ifdef DASP_EXPERIMENT      // breakdown of the step kernel (tools/mg_step_probe.py): the own-column part alone, inside the step kernel
    if (const char *e = std::getenv("DASP_MG_STEP_NOOTHER")) if (std::atoi(e)) { c.grid_b = 0; other = nullptr; }
#endif
    c.n_poll = std::min(c.grid_b, std::max(1, h.max_pollers));
    c.mark = static_cast<const unsigned char *>(h.mark); c.mark_members = static_cast<const unsigned *>(h.mark_members);
    c.n_marked = other ? h.n_marked : 0; c.n_mark_shards = h.n_mark_shards;
    c.blk_order = static_cast<const int *>(h.blk_order);
    c.sleep = std::max(1, h.poll_sleep); c.timeout = h.timeout_ticks;
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
    (void)hipGetLastError();
    return std::min(a, b);
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

}  // namespace dasp

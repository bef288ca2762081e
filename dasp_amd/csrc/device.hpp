// device.hpp -- device-side view of a plan, shared by kernels.hip (SpMV) and devpack.hip (packing on the GPU)
#pragma once

#include <cstddef>
#include <vector>

#include "plan.hpp"

namespace dasp {

struct ShortDev {
    int len, count, tiles, tile0;
    long long elem_off;
    SlotMap map;
    int seg, rpt;        // ShortGroup::seg / rpt
};

struct DevArgs {
    const void *x;
    void *y;
    // long
    const void *long_val; const int *long_cid; const int *piece_ptr; const int *piece_dst; void *partial;
    const int *multi_ptr; const int *multi_dst;
    int n_pieces, n_multi;
    // medium
    int pair_mode;                                           // Plan::pair_mode (plan.hpp med_npair)
    const int *med_ptr; const void *med_val; const int *med_cid;
    const unsigned short *med_cid16; const int *med_base;   // cid16 mode: u16 offsets + per-chunk base column
    const unsigned char *med_cid8; const int *med_c8ptr;     // ... and the one-byte offsets of the narrow chunks (plan.hpp med_cid8)
    const int *irr_ptr; const void *irr_val; const int *irr_cid;
    int n_blocks, row_block, row_long;
    // windowed mode (LDS-staged x)
    const int *med_dst; const int *win_cmin; const int *win_len;
    int n_windows, blocks_per_win;
    int win_hybrid;   // windows stage their densest span only: a gather outside [cmin, cmin + len) reads global memory
    int win_xcd;      // windows dealt to the XCDs in contiguous eighths (kernel)
    int win_rel16;    // 16-bit ids of an LDS-staged window are offsets from the window's first staged column (Plan::win_rel16)
    int ywt;   // 1: y stores written through (f64 plans whose tiles stream from HBM; put_y)
    int skip0; // 1: the rows without nonzeros are not stored at all (column panels: their slots of the partial-result buffer stay 0 from upload on)
    int acc;   // 1: y += A x (every y index has exactly one writer per launch, so a plain read-modify-write is exact)
    int wpw;   // waves per workgroup of this launch (4, or blocks_per_win in windowed mode)
    // short
    const void *short_val; const int *short_cid; const ShortDev *groups;
    int n_short_tiles;
    // short tiles per wave: 1, or kShortTpw consecutive tiles of a wave-segmented group in plans that have such groups (spmv_device.hpp short_waves; upload_plan sizes wg_short
    // with it).  Then wave w of the short range serves group gi = the last with grp_wave0[gi] <= w: tiles kShortTpw * (w - grp_wave0[gi]) .. of a segmented group, tile w - grp_wave0[gi] of any other
    int short_tpw, n_short_waves;
    int grp_wave0[kNumShortGroups];
    int grp_tile0[kNumShortGroups];   // first tile of every slab group (kernel arguments: the lookup is scalar)
    // permutation (DASP_Y_NATURAL only)
    const int *order;
    // workgroup ranges
    int wg_long, wg_med, wg_short;
    // medium blocks dealt to the 8 XCDs in contiguous ranges of equal work (upload_plan): workgroup m of the medium range serves blocks
    // xcd_blk[m % 8] + 4 * (m / 8) .. + 3 below xcd_blk[m % 8 + 1].  xcd_on = 0: block = workgroup * 4 + wave as ever.
    int xcd_on;
    int xcd_blk[9];
    int med_stride;   // 1: the medium workgroups stride over the blocks (capped, persistent range); 0: exactly one block per wave
    // row tiles of a column panel (Plan::rt_*): workgroups [wg_long + wg_med + wg_short, + wg_rt), one tile per wave
    const void *rt_val; const int *rt_cid; const int *rt_ptr; const unsigned short *rt_start; const unsigned long long *rt_mask;
    int n_rt_tiles, wg_rt, rt_max;
    // (r6; last, so that every older field keeps its offset: the multi-GPU step kernels take this block by value and their code depends on its layout)
    const unsigned short *long_cid16; const int *long_base; const int *piece_c16;      // 16-bit ids of the narrow long pieces (plan.hpp long_cid16)
    const int *med_nt;                                     // [blocks] tail steps of every medium block = ceil(tail entries of its first, longest row / entries per step) -- derived from irr_ptr at upload (r6)
    int wg_rot;       // workgroup b of the launch serves virtual workgroup (b + wg_rot) mod grid of the [long | medium | short] ranges (upload_plan: which category is dispatched first)
    int win_tiles;    // short tiles folded into every window workgroup (upload_plan): workgroup w also serves tiles w, w + n_windows, ... ; 0: the short tiles keep workgroups of their own
};

// two-phase form (plan.hpp struct TwoPhase): what its two kernels read.  All device pointers into the plan's arena; xs is the stream phase 1
// writes and phase 2 reads (one x value per stored element)
struct TpDev {
    const unsigned short *lcol; const int *dst; const int *unit;
    const void *val; const unsigned short *lrow; void *xs;
    const int *rb_row0; const int *rb_seg0;
    int n_units, n_rb, cb, rb_max, xlen, m;
};

// column-blocked long rows of a column-panel parent (plan.hpp struct LongCB): device pointers into the parent's arena
struct LcbDev {
    const void *val; const unsigned short *lcol; const int *ptr; const int *unit; const int *row_dst; void *partial;
    int n_units, n_rows, n_cb, cb, xlen;
};

// byte offsets of the nnz-sized arrays inside the arena (devpack.hip writes them, tests download them)
struct ArenaMap {
    size_t long_val = 0, long_cid = 0, med_val = 0, med_cid = 0, med_cid16 = 0, med_cid8 = 0, med_base = 0, irr_val = 0, irr_cid = 0,
           short_val = 0, short_cid = 0, rt_val = 0, rt_cid = 0, long_cid16 = 0, long_base = 0, piece_c16 = 0;
};

struct DevicePlan {
    // the plan's DevArgs as the kernels read it: a device-resident copy (x / y / acc / ywt unset -- those four travel in the kernarg segment), read
    // through the constant address space with scalar loads.  A 528-byte by-value block costs a small launch ~0.7 us more than a pointer to it
    // (tools/micro/launch_floor2.hip: 212 x 1024 threads, 3.8-3.95 vs 3.07-3.14 us per launch).  args_sent = what dargs currently holds:
    // launch_spmv re-sends the block when `args` was changed behind it (placement trials, the device packers).
    void *dargs = nullptr;
    void *args_sent = nullptr;      // host copy, sizeof(DevArgs)
    void *arena = nullptr;
    size_t arena_bytes = 0;
    ArenaMap map{};
    DevArgs args{};
    TpDev tp{};             // Plan::two_phase: the arena holds the tile streams, `args` is unused
    LcbDev lcb{};           // a column-panel parent with column-blocked long rows: its arrays follow the partial-result buffers in the arena
    bool nt = false;
    bool win1 = false;      // windowed plan with at most one window workgroup per CU: launch dasp_spmv_win1_kernel
    // f64, no windows: most of the regular chunks sit in ONE-SHOT blocks (rows of <= 32: all of a block's loads in flight at once, a wave's life is two memory round
    // trips for 3-8 KB) -- such a plan runs the build held to 72 registers = 7 waves per SIMD (kernels.hip; r5: rows of 40 0.74 -> 0.83, rows of 17 0.65 -> 0.71 of the
    // roofline; the pipelined blocks of long medium rows lose 4 % in that build and keep the unconstrained one)
    bool seven_waves = false;
    // r6: >= 5 % of the plan's nonzeros sit in narrow long pieces (plan.hpp long_cid16) of a plain plan: launch the builds that read their 16-bit ids (kernels.hip L16)
    bool long16 = false;
    int device = -1;
    // column-panel parent: arena = the panels' partial results, panel k at ypart + k * ypart_stride elements
    size_t ypart_stride = 0;
};


int require_device();                  // upload.cpp: DASP_OK, or DASP_ERR_NO_DEVICE with the error text set
int upload_plan(Plan &p);
void choose_long16(Plan &p);           // upload.cpp
int sync_dev_args(Plan &p);            // upload.cpp: DevicePlan::dargs = DevicePlan::args (a memcmp when nothing changed; a blocking copy otherwise -- never inside a stream capture: upload and the placement trials leave it in sync)
int upload_plan_unpacked(Plan &p);     // for the device packers: arena + O(rows) arrays, no placement trials yet
// kernels.hip: one SpMV of an uploaded plan (asynchronous); what upload.cpp asks the kernels
int launch_spmv(Plan &p, const void *dX, void *dY, void *stream, bool accumulate);
int spmv_kernel_allow_full_lds(int precision, bool c16);
int tp_kernels_allow_lds();             // kernels.hip: the two-phase kernels may use up to 160 KiB of dynamic LDS (called at upload)
int spmv_kernel_f16_resident(bool c16);
int tune_placement(Plan &p, int trials, const void *dX, void *dY, double *ms_first, double *ms_kept);   // kernels.hip: placement trials of an uploaded, fully packed plan (trials <= 0: default)

// fused multi-GPU step (kernels.hip; driven by multigpu.cpp): all pointers are device pointers
// the words of the fused step live in ONE zero-initialised device block of kMgWordBytes: sharded arrival counters first (marked
// own-column workgroups at 0, all workgroups at 8192, their top counters at 16384 / +256), then the flags, each on a line of its own
constexpr size_t kMgWordGathered = 20480, kMgWordOwnGo = 20480 + 4096, kMgWordReady = 20480 + 8192, kMgWordErr = 20480 + 12288, kMgWordXcd = 40960, kMgWordBytes = 40960 + 2048;      // (Xcd: 8 x 256 bytes, the per-XCD acquire words)
struct MgStepCtl {
    void *words;                                       // the block above
    unsigned long long need;                           // the other-column product waits in the kernel for gathered >= need (0: no wait)
    unsigned long long step;                           // published as "ready" by the last workgroup of the launch
    const void *mark; const void *mark_members;        // device tables: [own workgroups] bytes, [64] marked workgroups per shard
    const void *blk_order;                             // device table: [own medium blocks] dispatch order
    int n_marked, n_mark_shards;
    int max_pollers;                                   // bound on the persistent workgroups that wait
    int poll_sleep;                                    // pause between two polls of a workgroup, in units of s_sleep(8) (~0.2 us)
    long long timeout_ticks;                           // 100 MHz ticks a wait may take before it gives up and sets the error word
    double poll_at;                                    // where the waiting workgroups stand in the grid, as a fraction of the own-column workgroups (1: last)
    void *err;                                         // the sticky error word (host-mapped, so that the host reads it without a synchronisation); null: words + kMgWordErr
    int ready_by_event;                                // 1: the kernel does not publish "ready"; the caller does, behind the launch on the same stream (multigpu.cpp ready_by_kernel)
};
bool mg_step_supported(const Plan &own, const Plan *other);
// host: which own-column workgroups of the step kernel store a row with has_other[row] != 0 (natural-order plan, before the
// host arrays are dropped).  mark gets one byte per workgroup of the plan's launch grid; blk_order the dispatch order of the medium
// blocks that the marks assume (blocks holding such rows first).
void mg_step_marks(const Plan &own, const unsigned char *has_other, std::vector<unsigned char> &mark, std::vector<int> &blk_order, double hot_at = -1.0);
int launch_mg_step(Plan &own, Plan *other, const void *x_own, const void *x_gathered, void *y, const MgStepCtl &c, void *stream);
int mg_step_resident_per_cu();
// the step on ONE stream (mgstep.hip dasp_mg_step2_kernel): ONE plan in the gather-buffer column layout; head workgroups send the previous
// slice to the peers (push: mgx.hpp, n_dst may be 0), unmarked workgroups run at once, the marked ones (boundary rows) behind every peer's
// arrival flag -- each by itself where it stands in the list, or (max_pollers > 0) through a bounded set of persistent workgroups at the end.
// All pointers are device pointers.
struct MgStep2Ctl {
    const void *wg_list; const void *blk_order;      // device tables: [n_free + n_marked] virtual workgroups (free first), [medium blocks] dispatch order
    int n_push, n_free, n_marked, n_total, max_pollers;   // list = n_free unmarked, n_marked marked, the remaining unmarked; max_pollers 0: marked workgroups wait in place
    const void *arrived; int world, rank;            // this rank's arrival flags
    unsigned long long need;                         // value the flags must reach (0: no wait)
    void *err; long long timeout_ticks; int poll_sleep;
    int fence_mode;                                  // 0: every waiting workgroup acquires at system scope; 1: one per XCD (default)
    void *xcd_fenced; unsigned long long step;       // device: 8 x 256 bytes, zeroed at upload; the launch's number (monotone)
};
struct MgPushArgs;
int launch_mg_step2(Plan &plan, const void *x, void *y, const MgStep2Ctl &c, const MgPushArgs &push, void *stream);
int launch_mg_wait(const void *word, unsigned long long need, long long timeout_ticks, void *err, void *stream);
int launch_mg_flag(void *word, unsigned long long value, void *stream);

}  // namespace dasp

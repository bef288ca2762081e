// device.hpp -- device-side view of a plan, shared by kernels.hip (SpMV) and devpack.hip (packing on the GPU)
#pragma once

#include <cstddef>

#include "plan.hpp"

namespace dasp {

struct ShortDev {
    int len, count, tiles, tile0;
    long long elem_off;
    SlotMap map;
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
    int acc;   // 1: y += A x (every y index has exactly one writer per launch, so a plain read-modify-write is exact)
    int wpw;   // waves per workgroup of this launch (4, or blocks_per_win in windowed mode)
    // short
    const void *short_val; const int *short_cid; const ShortDev *groups;
    int n_short_tiles;
    int grp_tile0[kNumShortGroups];   // first tile of every slab group (kernel arguments: the lookup is scalar)
    // permutation (DASP_Y_NATURAL only)
    const int *order;
    // workgroup ranges
    int wg_long, wg_med, wg_short;
};

// byte offsets of the nnz-sized arrays inside the arena (devpack.hip writes them, tests download them)
struct ArenaMap {
    size_t long_val = 0, long_cid = 0, med_val = 0, med_cid = 0, med_cid16 = 0, med_cid8 = 0, med_base = 0, irr_val = 0, irr_cid = 0,
           short_val = 0, short_cid = 0;
};

struct DevicePlan {
    void *arena = nullptr;
    size_t arena_bytes = 0;
    ArenaMap map{};
    DevArgs args{};
    bool nt = false;
    int device = -1;
    // column-panel parent: arena = the panels' partial results, panel k at ypart + k * ypart_stride elements
    size_t ypart_stride = 0;
};


int upload_plan(Plan &p);

}  // namespace dasp

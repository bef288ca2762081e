/*
 * dasp_amd.h -- C ABI of libdasp_amd.so: MI355X (gfx950) implementation of the DASP SpMV
 * hot path.  Plain pointers and sizes only; no C++/torch types cross this boundary.
 *
 * Every entry point cites the reference interface it replaces (paths relative to the
 * reference tree, SuperScientificSoftwareLaboratory/DASP).  The reference has no FFI layer:
 * its boundary is two C++ free functions (mmio_allinone, spmv_all) called from main()
 * (src/main_f64.cu:129,149; src/main_f16.cu:131,146) -- those are what a maintainer would
 * re-bind (INTEGRATION.md shows the stub).
 *
 * Conventions
 *   - all functions return 0 on success or a negative dasp_status (never abort, never print
 *     on error); the loader keeps the reference's -1/-2/-4 codes.
 *   - half precision values cross the boundary as IEEE binary16 bit patterns (uint16_t).
 *   - device pointers are HIP device pointers of the current process; `stream` is a
 *     hipStream_t passed as void* (NULL = the null stream).
 *   - the GPU path has NO CPU fallback: without a usable HIP device the device entry points
 *     return DASP_ERR_NO_DEVICE / DASP_ERR_HIP.
 */
#ifndef DASP_AMD_H
#define DASP_AMD_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef enum dasp_status {
    DASP_OK = 0,
    DASP_ERR_OPEN = -1,       /* mmio_allinone: fopen failed        (mmio_highlevel.h:623-624) */
    DASP_ERR_BANNER = -2,     /* mmio_allinone: bad banner          (mmio_highlevel.h:626-630) */
    DASP_ERR_SIZE = -4,       /* mmio_allinone: bad size line       (mmio_highlevel.h:638-640) */
    DASP_ERR_ENTRY = -5,      /* malformed / missing entry, index out of range (reference: undefined) */
    DASP_ERR_ARG = -10,
    DASP_ERR_NOMEM = -11,
    DASP_ERR_NO_DEVICE = -20, /* no HIP device visible */
    DASP_ERR_HIP = -21,       /* a HIP runtime call failed; see dasp_last_error() */
    DASP_ERR_STATE = -22      /* e.g. spmv before upload */
} dasp_status;

/* thread-local text of the last failure ("" if none) */
const char *dasp_last_error(void);
/* "dasp_amd <version> gfx950" */
const char *dasp_version(void);

/* ------------------------------------------------------------------ loader
 * Replaces  int mmio_allinone(int *m, int *n, MAT_PTR_TYPE *nnz, int *isSymmetric,
 *                             MAT_PTR_TYPE **csrRowPtr, int **csrColIdx,
 *                             MAT_VAL_TYPE **csrVal, char *filename)
 * (src/mmio_highlevel.h:608-610; banner/size rules src/mmio.h:398-624).
 * Same out-parameters, same CSR (file order inside a row, symmetric/hermitian mirrored
 * right after the source entry, duplicates kept, columns not sorted), same return codes.
 * The three arrays are malloc'd; release each with dasp_free (the reference's caller uses
 * free(): src/main_f64.cu:162-164). */
int dasp_mmio_allinone_f64(int *m, int *n, int *nnz, int *isSymmetric,
                           int **csrRowPtr, int **csrColIdx, double **csrVal, const char *filename);
int dasp_mmio_allinone_f16(int *m, int *n, int *nnz, int *isSymmetric,
                           int **csrRowPtr, int **csrColIdx, uint16_t **csrVal, const char *filename);
void dasp_free(void *p);

/* binary cache of the loader's CSR (the reference re-parses the text file on every run).
 * dasp_csr_load returns exactly what dasp_mmio_allinone_* returned when the file was saved; csrVal is double* or
 * uint16_t* (binary16) according to `precision`; arrays are malloc'd (dasp_free). */
int dasp_csr_save(const char *path, int precision, int m, int n, int nnz, int isSymmetric,
                  const int *csrRowPtr, const int *csrColIdx, const void *csrVal);
int dasp_csr_load(const char *path, int precision, int *m, int *n, int *nnz, int *isSymmetric,
                  int **csrRowPtr, int **csrColIdx, void **csrVal);

/* --------------------------------------------------------------- plan API
 * The reference fuses preprocessing, upload, 1100 launches and download in spmv_all
 * (src/dasp_f64.h:486-1483).  The plan API is the same work split into reusable steps. */
typedef struct dasp_plan dasp_plan_t;

typedef enum { DASP_F64 = 64, DASP_F16 = 16 } dasp_precision;
typedef enum {
    DASP_Y_PERMUTED = 0, /* y[i] belongs to row order_rid[i]: the reference's output (dasp_f64.h:1402) */
    DASP_Y_NATURAL = 1   /* y[r] belongs to row r: un-permute fused into the kernel's stores */
} dasp_y_order;

typedef struct dasp_options {
    double threshold;      /* regular/irregular fill threshold; reference: 0.75 (main_f64.cu:125) */
    int block_longest;     /* rows with >= this many nonzeros are "long"; reference: 256 (main_f64.cu:124) */
    int y_order;           /* dasp_y_order */
    int long_piece;        /* nonzeros of a long row given to one wave; 0 = default (1024; the longest row when the long
                              rows hold <= 2 M nonzeros in rows of <= 16384, or when no row is longer than 4096, so that no second launch is needed --
                              unless one wave walking the longest row would outlast the rest of the launch: ~0.35 us per 256 f64 / 512 f16 elements
                              against ~7 us + the matrix at 5 TB/s; then pieces of 1024 again) */
    int host_threads;      /* preprocessing threads; 0 = hardware concurrency */
    /* column remap for the row-partitioned multi-GPU layout (0/NULL = identity):
     * column c owned by part g (part_bounds[g] <= c < part_bounds[g+1]) is read from
     * x[g * part_stride + (c - part_bounds[g])], i.e. straight out of an all-gather buffer
     * of equal-size padded slices. */
    int n_parts;
    const int *part_bounds; /* [n_parts+1] */
    int part_stride;
    /* LDS-staged x gathers.  Medium rows are blocked inside windows of `row_window` consecutive rows (sorted by
     * length inside the window only) so that one workgroup's rows share a narrow span of x; that span is copied
     * once into LDS (coalesced) and every gather of the window is served from LDS.  y still goes to the slots
     * of the reference permutation (order_rid is unchanged), through a per-row destination table.
     *   x_window: 0 = auto (on when the windows of >= half of the medium nonzeros fit AND the rows' gathers are scattered:
     *             >= 60 % of a sampled row's nonzeros lie on another 128-byte line of x than their predecessor -- or, from 16 M nonzeros on,
     *             when it is the global length sort that scatters: sampled blocks of the sorted order hold rows more than 16 apart), -1 = off,
     *             -2 = windowed order without LDS staging (measurement knob: slower than either alternative),
     *             > 0 = force on with this many bytes of LDS per workgroup as the cap (<= 163840; auto uses 81920,
     *                   i.e. two workgroups per CU, and falls back to 163840 when the spans do not fit that)
     *   row_window: rows per window, multiple of 16 (one block) from 64 up to 1024 (up to 16 waves per workgroup, 1-4 blocks per wave);
     *               0 = by size: as many windows as put a workgroup on every CU and none behind another (windows rounded up to 8 + long + short workgroups <= 256; r6),
     *               ~448 (two per CU) once a window would pass 1024 rows */
    int x_window;
    int row_window;
    /* 16-bit column ids for the regular medium tiles: u16 offsets from a per-chunk base column (10 instead of 12 bytes
     * per f64 nonzero, 4 instead of 6 for f16).  Chunks spanning more than 65534 columns end their block's regular
     * part (the rest goes to the 32-bit irregular tail).  0 = auto (on for >= 64 MiB of CSR without x windows -- 256 MiB with them -- when that loses < 3 % of the regular elements),
     * -1 = off, 1 = force on. */
    int cid16;
    /* cache policy of the streamed tiles: 0 = auto (non-temporal when the packed matrix exceeds 256 MiB and cannot stay in
     * the Infinity Cache anyway), 1 = plain loads (the reference's dasp_spmv, dasp_f16.h:593-1013), 2 = non-temporal loads
     * (the reference's "bypass" kernel dasp_spmv2 with ld.global.cs, dasp_f64.h:34-51).  x gathers always use plain loads. */
    int stream_policy;
    /* column panels (cache blocking for gather-bound matrices, no reference counterpart).  The matrix is split into
     * `col_panels` column ranges of equal width, each packed as its own DASP plan and launched back to back, so that
     * the x entries a launch gathers (x_len / col_panels of them) stay in the 4 MiB L2 of every XCD; a last streaming
     * kernel adds the panels' partial results.  order_rid and the classifier counters stay those of the whole matrix.
     *   0 = auto (on for matrices whose rows scatter over more x than the L2 holds: x > 4 MiB, >= 16 M nonzeros,
     *       > 75 % of a row's nonzeros on distinct 128-byte lines of x, rows spanning > x/4, and < 80 % of the gathers on the
     *       hottest 3 MiB of x lines; for a device-resident CSR the samples are gathered by a kernel and the split runs on the GPU;
     *       also 2 panels when x fits the L2 but rows long enough for long_cb hold half the nonzeros: the panel plan is where they live),
     *   1 or -1 = off,  2..64 = that many panels.
     * f16: the per-panel partial results are rounded to binary16 before they are added (in f32). */
    int col_panels;
    /* medium rows of at most this many nonzeros are stored as uniform-length slabs (one slab per length, a lane owns whole rows,
     * no MFMA) like the reference's short rows, instead of 16-row MFMA blocks: a block of rows that short is mostly per-wave
     * overhead (8 M rows of 5 nonzeros: 0.39 of the roofline as blocks, 0.8+ as slabs).  Their slots in order_rid stay the
     * medium rows' (sorted by length).  0 = auto: up to 16 (f64) / 24 (f16) nonzeros, and only when neighbouring rows read
     * neighbouring columns (stencils: the lanes' gathers coalesce; on graph-like rows slabs lose) and those
     * rows hold at least half of the nonzeros ("neighbouring" = within a 128-byte line of x; within 512 columns for f16 candidates that are mostly rows
     * of fewer than 12 nonzeros: an f16 tile holds 16 columns of a row, such rows never fill a regular chunk and their blocks are all tail steps);
     * 4 = off; 5..32 = that bound, unconditionally. */
    int slab_max_len;
    /* hybrid x windows (graph-like rows: most columns near the rows, a scattered remainder).  When the whole span of a window
     * does not fit in LDS, the window's DENSEST span of `x_window` bytes (auto: 81920) is staged and the gathers that fall outside
     * it read global memory.  0 = auto: when the strict windows cover < half of the medium nonzeros, the densest spans would
     * cover >= 60 % and the rows are of even length (longest <= 4 x the mean: a band plus outliers; on power-law rows the window
     * workgroups cost more than the LDS gathers save); -1 = never; 1 = force (with x_window >= 0).  Host CSR only. */
    int x_window_hybrid;
    /* the longest medium rows as wave-sized pieces (stored and multiplied like the long rows: CSR order, one wave per piece of
     * <= long_piece nonzeros, all 64 lanes on one row) instead of 16-row blocks, whose ceil(len / K) MFMA steps run serially in one
     * wave.  Their slots in order_rid and every classifier counter stay the medium rows'.  0 = auto: only on latency-bound
     * matrices (two batches of the longest block would already take as long as streaming the matrix) whose long medium rows are
     * a tail (<= 15 % of the nonzeros); -1 = off; n >= 5 = every medium row of >= n nonzeros.  Ignored by plans that use x windows
     * (their rows keep the LDS gathers). */
    int piece_min_len;
    /* paired medium chunks: blocks store two MFMA steps per lane side by side so that one 16-byte load brings both (half the L1 tag
     * lookups of the streamed tiles).  0 = auto (1, or 2 for plans of more than 1 GiB of CSR; LDS-windowed plans: off);
     * -1 = off; 1 = the blocks long enough for the kernel's software pipeline and the one-shot f16 blocks; 2 = also the one-shot f64 blocks
     * that have no tail steps.  order_rid, the classifier counters and the arithmetic of a row do not depend on it. */
    int chunk_pairs;
    /* one-byte column ids (f64, inside cid16 mode, pipelined paired chunks only): a chunk whose columns span <= 254 stores its offsets from
     * the chunk's base column in one byte; such chunks are moved to the front of their block's paired region in whole pipeline batches
     * (the order of a block's MFMA steps is free), a batch's ids being one dword per lane.  0 = auto: in pipelined blocks only, and only when at least a tenth of the plan's
     * chunks qualify (one-shot blocks are faster with 16-bit ids, and a plan with any one-byte id runs its own, 80-register kernel build); 1 = wherever it applies, one-shot blocks
     * included (whole pairs); -1 = off. */
    int cid8;
    /* wave-segmented short rows (BASELINE north_star; reference branches dasp_f64.h:281-483): rows of 1..4 nonzeros stored back to back, ONE
     * nonzero per lane, every 16-lane DPP row holding 16 / L whole rows; a row is summed with two DPP row_shl steps and the lane holding its
     * first nonzero stores y -- instead of uniform-length slabs in which a lane owns whole rows (no cross-lane step, 16-byte loads, half to a
     * quarter as many waves).  0 = auto: on for f64 plans without x windows (same-device A/B: webbase-1M f64 30.6 -> 28.4 us, powerlaw_1M
     * 0.668 -> 0.662 ms; f16 and the windowed kernels lose); -1 = off (slabs); 1 = on.  order_rid, every counter
     * and the order of a row's products are unchanged; the sum of a row of 4 is (p0 + p1) + (p2 + p3) instead of ((p0 + p1) + p2) + p3. */
    int short_seg;
    /* row tiles of the column panels (no reference counterpart): a panel's rows of at most `row_tile_max` nonzeros IN THAT
     * PANEL are not classified and sorted by the panel but kept in the parent's output order, in tiles of 64 consecutive output positions: one
     * wave streams a tile's (value, column) pairs in CSR order, parks the products in LDS, lane r sums row r's products in their CSR order and
     * the wave stores ONE complete 128-byte line of the panel's partial result (f16; two lines in f64) -- instead of 2-byte pieces of lines
     * that other workgroups complete (ljournal-2008 f16: 333 MB written per SpMV for 54 MB of partial results).  The rows above the bound stay
     * with the panel's own blocks / pieces.  0 = auto (f16: 16 -- ljournal-2008 0.507 -> 0.436 ms, 333 -> 69 MB written; f64: 8 -- powerlaw_1M 0.653 -> 0.648,
     * 16 KB of LDS per workgroup so that a CU still holds six of them); -1 = off; 1..32 = that bound (LDS: 256 x bound products per workgroup).  Plans without column panels ignore it.  order_rid, the
     * classifier counters and the order of a row's products are unchanged (a tiled row is summed in CSR order, as the oracle does). */
    int row_tile_max;
    /* rows whose column ids do not ascend (a CSR assembled from unsorted coordinates): the packers keep the order of a row's entries, as the reference does, and the
     * gathers of such rows touch more lines of x (graph-like rows: 13-16 % slower, profiles/r04_row_tiles.md 14).  1 = sort every such row's (column, value) pairs by
     * column before packing -- stable, on the host or (dasp_plan_create_device) with a segmented sort on the GPU, bit-identical either way; the caller's arrays are not
     * touched.  The sum of a row then runs in column order instead of CSR order (same products).  0 / -1 = keep the CSR order (default: what the reference does). */
    int sort_columns;
    /* two-phase (gather-free) form, f16 plans only (no reference counterpart).  For matrices whose rows scatter over all of x every
     * gathering kernel pays ~0.8 L1 misses per nonzero.  Here the nonzeros are cut into (row block, column block) tiles; phase 1 stages a column block's
     * slice of x in LDS and expands it into a stream xs of one x value per nonzero, phase 2 keeps a row block's slice of y in LDS (f64 accumulators) and
     * streams (value, local row, xs): every byte is streamed, 10 B per nonzero against the 6 of B_alg, nothing gathers from global memory
     * (ljournal-2008 f16: 0.43 -> 0.22 ms, the uniform-column variant 0.51 -> 0.18).  All rows take this path; order_rid and the classifier counters
     * stay those of the whole matrix; products are f16 x f16 accumulated in f64 (the order of a row's additions is not fixed: LDS atomics).
     *   0 = auto: f16, no column remap, no explicit col_panels, >= 10 M nonzeros whose rows scatter (> 50 % of a sampled row's nonzeros on distinct 128-byte lines of x,
     *       a third of the entries in rows spanning > x/4; hub rows are fine: same-row elements are combined before they reach LDS); 1 = force; -1 = off.
     *       The automatic rule also declines (the matrix keeps its DASP form) when the tiles' padding to whole 64-element segments would store > 3 x the nonzeros
     *       (large, very sparse matrices: few nonzeros per tile) or the tile table would pass 64 M entries.
     *   DETERMINISM: this is the one form whose results are not bit-reproducible from run to run -- a row's products reach its f64 LDS accumulator through relaxed
     *   atomics, so their order of addition is not fixed (the f64 sum rounds ~2^-53; the difference shows only where the final rounding to f16 sits on a tie).  Every other
     *   path (DASP blocks, slabs, panels, column-blocked long rows, the multi-GPU step) adds in a fixed order.  two_phase = -1 is the deterministic choice.
     *   tp_col_block: columns per column block (multiple of 8, <= 65536; 0 = 32768); tp_row_block: most output positions per row block (<= 8192; 0 = 4096). */
    int two_phase;
    int tp_col_block, tp_row_block;
    /* column-blocked long rows of a column-panel plan (no reference counterpart; the north_star's LDS-staged x gathers applied to the long rows): the rows of
     * >= max(block_longest, 64 x column blocks) nonzeros leave the panels; their nonzeros are cut by column block (16384 columns in f64, 32768 in f16: the block's
     * slice of x staged in LDS), one wave per (row, block) piece multiplies value x LDS-x, a second kernel adds a row's partials.  value + u16 local column are
     * streamed, nothing gathers from global memory.  0 = auto (when those rows hold >= a quarter of the nonzeros of a plan that uses column panels), 1 = force
     * (every row of >= block_longest nonzeros), -1 = off.  order_rid and the classifier counters are unchanged; a long row's products are added per piece. */
    int long_cb;
} dasp_options_t;

void dasp_options_default(dasp_options_t *opt);

/* the reference's CSV counters (src/dasp_f64.h:1439-1441) + this build's own layout sizes */
typedef struct dasp_stats {
    int precision, rowA, colA, nnzA;
    /* classifier -- identical to the reference's columns (short_row_1/3 after pairing) */
    int short_row_1, common_13, short_row_3, short_row_4, short_row_2, row_long, row_block, row_zero;
    int nnz_short, nnz_long, origin_nnz_reg, nnz_irreg;
    int rowloop;               /* reference's 59990/400000 rule, reported only (dasp_f64.h:533-536) */
    /* native (gfx950 geometry) padded sizes, the analogue of fill0_nnz_*; fill0_nnz_short counts the whole slab segment, i.e. also
     * the medium rows stored as slabs (slab_max_len), fill0_nnz_reg / n_med_blocks only the MFMA blocks */
    long long fill0_nnz_short, fill0_nnz_long, fill0_nnz_reg;
    double rate_fill0;         /* (stored slots - nnzA) / nnzA, as dasp_f64.h:1159-1160 */
    long long data_X;          /* bytes of the packed format + x + y, as dasp_f64.h:1162-1166 */
    long long data_origin1;    /* CSR algorithmic bytes, main_f64.cu:143 */
    int n_med_blocks, n_long_pieces, n_long_multi, n_short_tiles, n_workgroups;
    double pre_ms;             /* host preprocessing wall time (dasp_f16.h:1444-1445 "dasp_pre") */
    /* LDS-staged x windows (0 everywhere when the mode is off) */
    int x_window_on, n_windows, n_windows_lds, lds_bytes, row_window;
    double window_nnz_frac;    /* share of the medium nonzeros whose window fits in LDS */
    int cid16_on;              /* regular medium tiles carry 16-bit column ids */
    int n_col_panels;          /* 0 = single plan; else the number of (non-empty) column panels: the fill0_*, data_X, n_* and
                                  window / cid16 fields are then sums (or any-of) over the panels */
    int x_window_hybrid;       /* windows stage their densest span; window_nnz_frac = share of the medium gathers served from LDS */
    int med_rows_as_pieces;    /* medium rows (the longest ones: the first medium slots) stored as pieces (piece_min_len) */
    int chunk_pairs;           /* 0 / 1 / 2: which medium blocks store chunk pairs (options chunk_pairs; column panels: the largest) */
    int cid8_chunks;           /* regular medium chunks with one-byte column ids (option cid8) */
    int short_seg;             /* 1: the short rows (1..4 nonzeros) use the wave-segmented DPP layout */
    int row_tile_max;          /* column panels: rows of at most this many nonzeros per panel are stored as row tiles (0: none) */
    int n_row_tiles;           /* row tiles over all panels */
    long long row_tile_nnz;    /* nonzeros stored in row tiles */
    int two_phase;             /* 1: the plan is in the two-phase (gather-free) form (option two_phase) */
    int tp_col_block, tp_row_blocks, tp_units;   /* its columns per column block, row blocks (phase-2 workgroups) and phase-1 workgroups */
    long long tp_segments;     /* its segments: stored (padded) elements / tp_seg_elems */
    int tp_seg_elems;          /* elements per segment (64) */
    int lcb_rows, lcb_col_block, lcb_units;   /* column-blocked long rows of a column-panel plan (option long_cb): rows taken, columns per block, workgroups */
    long long lcb_elems;       /* their stored (padded) elements */
    /* the REFERENCE's geometry on the same input (8-row blocks, 8x4 tiles, 32-lane warps): the padded sizes the CUDA reference computes
     * and writes into its CSV row for this matrix -- short tiles dasp_f64.h:609-629 / dasp_f16.h:1139-1156, long rows :1000-1014 /
     * :1273-1288, regular / irregular split :1044-1091 / :1317-1365, rate_fill0 and data_X :1159-1166 / dasp_f16.h:1448-1455.  Functions
     * of the row lengths alone; dasp_spmv_all_* writes THESE into the reference's CSV columns, so that a row written here can be diffed
     * against a row the reference writes (the native sizes above go to data/dasp_amd_native_f64.csv / _f16.csv). */
    long long ref_fill0_nnz_short, ref_fill0_nnz_long, ref_fill0_nnz_reg, ref_data_X;
    int ref_nnz_irreg, ref_origin_nnz_reg, ref_blocknum, ref_warp_number;
    double ref_rate_fill0;
} dasp_stats_t;

/* classifier + packers on the host (no GPU needed).  CSR arrays are read-only and may be
 * freed afterwards.  csrVal: double* for DASP_F64, uint16_t* (binary16) for DASP_F16.
 * Mirrors the host part of spmv_all: dasp_f64.h:499-1157 / dasp_f16.h:1029-1443. */
int dasp_plan_create(dasp_plan_t **plan, int precision, int rowA, int colA, int nnzA,
                     const int *csrRowPtr, const int *csrColIdx, const void *csrVal,
                     const dasp_options_t *opt /* NULL = defaults */);
/* the same with the CSR already on the current HIP device (dRowPtr / dColIdx / dVal are device pointers): only the row
 * pointer visits the host; the nonzeros are range-checked, scanned, split into column panels where the plan uses them, and packed by
 * kernels.  The plan comes back uploaded and produces bit-identical packed arrays to dasp_plan_create, with the same
 * automatic choices (only hybrid x windows are decided on a host CSR alone). */
int dasp_plan_create_device(dasp_plan_t **plan, int precision, int rowA, int colA, int nnzA,
                            const int *dRowPtr, const int *dColIdx, const void *dVal, const dasp_options_t *opt);
/* Placement trials (r3; opt-in since r4): the same packed bytes run the HBM-bound kernels at one of two speeds ~8 % apart depending on where the
 * plan's device allocation and the written vector y landed relative to each other (profiles/r03_placement.md, r04_placement.md).  Copies the arena
 * into up to `trials` - 1 fresh allocations (<= 0: 2, i.e. ONE extra allocation; never when the device has less free memory than the arena + 1 GiB),
 * times each with dX / dY and keeps the fastest; the others are freed before the call returns.  No sleeps, no launches behind the trials: the driver
 * wipes released VRAM in the background (~35 GB/s) and kernels run 1-3 % slower meanwhile.  Only plans that stream >= 256 MiB and have no x windows
 * / column panels; others return at once.  dasp_plan_upload does NOT run it (r3 did) unless DASP_PLACEMENT_TRIALS=n > 1 is set; the one-shot
 * dasp_spmv_all_* never does.
 * dX / dY: the caller's own device vectors (x_len / rowA elements; dY is overwritten), or NULL for scratch ones -- it is y's placement against the
 * arena's that decides, so a solver that keeps its vectors should lend them (or simply try a few y allocations of its own: 16 MB each instead of a
 * copy of the plan).  ms_first / ms_kept (may be NULL): the time of the first and of the kept allocation, 0 when nothing was tried.  Synchronises the
 * device; results of later products are unchanged (same bytes). */
int dasp_plan_tune_placement(dasp_plan_t *plan, int trials, const void *dX, void *dY, double *ms_first, double *ms_kept);
/* copy one nnz-sized packed array (names as dasp_plan_host_array) from the device arena to `dst` (tests, serialisation) */
int dasp_plan_download_array(dasp_plan_t *plan, const char *name, void *dst, size_t bytes);
void dasp_plan_destroy(dasp_plan_t *plan);

/* serialised plan (no reference counterpart): every packed host array + order_rid + stats, so that a later
 * run or another rank goes load -> upload -> spmv without re-packing.  Saving needs the host arrays (before
 * dasp_plan_drop_host). */
int dasp_plan_save(dasp_plan_t *plan, const char *path);
int dasp_plan_load(dasp_plan_t **plan, const char *path);

/* order_rid[i] = original row of permuted slot i (dasp_f64.h:960-976 / dasp_f16.h:1253-1270);
 * identical to the reference's array.  Owned by the plan. */
const int *dasp_plan_order(const dasp_plan_t *plan);
int dasp_plan_stats(const dasp_plan_t *plan, dasp_stats_t *out);
/* dasp_y_order of the plan; number of x elements a SpMV reads (colA, or n_parts * part_stride in the partitioned layout) */
int dasp_plan_y_order(const dasp_plan_t *plan);
long long dasp_plan_x_len(const dasp_plan_t *plan);

/* column panels of a plan built with col_panels (0 for a single plan), and a BORROWED handle to panel k: a natural-order
 * plan over the same rows whose column ids lie in the panel's range.  The handle belongs to the parent (never destroy it);
 * it answers the query functions (stats, host_array, ...) and dasp_plan_panel_range. */
int dasp_plan_panel_count(const dasp_plan_t *plan);
dasp_plan_t *dasp_plan_panel(dasp_plan_t *plan, int k);
int dasp_plan_panel_range(const dasp_plan_t *plan, int k, int *col_begin, int *col_end);

/* read-only view of a packed host array, for format tests and serialisation.
 * returns element count, or a negative dasp_status for an unknown name. */
long long dasp_plan_host_array(const dasp_plan_t *plan, const char *name, const void **ptr, int *elem_bytes);

/* hipMalloc + H2D of the packed arrays on the CURRENT device (dasp_f64.h:1239-1278) */
int dasp_plan_upload(dasp_plan_t *plan);
/* release the host copies of the packed arrays once uploaded (order_rid and stats stay) */
int dasp_plan_drop_host(dasp_plan_t *plan);

/* switch the cache policy of an uploaded plan (values as dasp_options_t::stream_policy); no re-upload */
int dasp_plan_set_stream_policy(dasp_plan_t *plan, int policy);

/* one SpMV, y = A*x, asynchronous on `stream`.  dX: colA values (or the part_stride layout),
 * dY: rowA values, both device pointers of the plan's precision (dX 16-byte aligned when the plan uses x windows).
 * Replaces the launches dasp_spmv2<rowloop><<<>>> + longPart_sum<<<>>>
 * (dasp_f64.h:1291-1319 / dasp_f16.h:1548-1704).  Only kernel launches: safe inside a hipStreamBeginCapture region.
 * One SpMV of a given plan may be in flight at a time (rows cut into several pieces share the plan's partial-sum
 * buffer); different plans are independent. */
int dasp_plan_spmv(dasp_plan_t *plan, const void *dX, void *dY, void *stream);
/* the same launches with y += A*x: every y index has exactly one writer per launch, so the update is a plain
 * read-modify-write (deterministic; f16: old y widened to f32, added, rounded once).  Lets a matrix split by columns
 * into several plans (dasp_mg_spmv: own / other ranks' columns) produce one y without a separate add. */
int dasp_plan_spmv_acc(dasp_plan_t *plan, const void *dX, void *dY, void *stream);

/* the reference's timing protocol (dasp_f64.h:1285-1320,1394): `warmup` untimed + `iters`
 * timed back-to-back SpMVs on `stream`, one sync at the end.
 * wall_ms_per_iter: host clock; event_ms_per_iter: hipEvent pair recorded on `stream`. */
int dasp_plan_time(dasp_plan_t *plan, const void *dX, void *dY, void *stream, int warmup, int iters,
                   double *wall_ms_per_iter, double *event_ms_per_iter);

/* `iters` back-to-back SpMVs with a hipEvent between every two of them: ms_each[i] = event i -> event i + 1, i.e. launch i's duration
 * plus the gap to its successor (a few us).  For the spread of a kernel's duration inside one process (min / median / max). */
int dasp_plan_time_each(dasp_plan_t *plan, const void *dX, void *dY, void *stream, int warmup, int iters, float *ms_each);

/* the same protocol with `batch` SpMVs captured once into a hipGraph and replayed ceil(iters/batch) times:
 * removes the per-launch host cost that bounds back-to-back launches on small matrices (kernels unchanged).
 * stream NULL = a private capture stream. */
int dasp_plan_time_graph(dasp_plan_t *plan, const void *dX, void *dY, void *stream, int warmup, int iters, int batch,
                         double *wall_ms_per_iter, double *event_ms_per_iter);

/* ---------------------------------------------------------------- one-shot
 * Replaces  void spmv_all(char *filename, MAT_VAL_TYPE *csrValA, MAT_PTR_TYPE *csrRowPtrA,
 *                         int *csrColIdxA, MAT_VAL_TYPE *X_val, MAT_VAL_TYPE *Y_val,
 *                         int *order_rid, int rowA, int colA, MAT_PTR_TYPE nnzA, int NUM,
 *                         double threshold, int block_longest)
 * (src/dasp_f64.h:486-487, src/dasp_f16.h:1015-1016).  Host buffers in, Y_val (permuted) and
 * order_rid out; prints the reference's "SpMV_X: ms, GFlop/s, GB/s, GB/s" line and appends
 * the reference's CSV row to data/spmv_f64_record.csv / data/spmv_f16_record.csv when the
 * data/ directory exists.  NUM is accepted and ignored, as in the reference. */
int dasp_spmv_all_f64(const char *filename, const double *csrValA, const int *csrRowPtrA,
                      const int *csrColIdxA, const double *X_val, double *Y_val, int *order_rid,
                      int rowA, int colA, int nnzA, int NUM, double threshold, int block_longest);
int dasp_spmv_all_f16(const char *filename, const uint16_t *csrValA, const int *csrRowPtrA,
                      const int *csrColIdxA, const uint16_t *X_val, uint16_t *Y_val, int *order_rid,
                      int rowA, int colA, int nnzA, int NUM, double threshold, int block_longest);

/* ---------------------------------------------------------------- multi-GPU
 * No reference counterpart: the reference drives one device (src/main_f64.cu:102-168).  Row-partitioned y = A*x over the
 * GPUs of one node, ONE PROCESS PER GPU: rank g owns the contiguous rows [row_bounds[g], row_bounds[g+1]) (and, for a square
 * matrix, the x entries of the same indices), runs the complete DASP pipeline on its slice and sends its y slice to every
 * rank with one RCCL all-gather over xGMI; the gathered y is laid out exactly as the next product reads x
 * (x_{t+1} = y_t, what an iterative solver does), in equal padded slices: element i of rank g's slice sits at
 * g * stride + i.  With `overlap` the slice is split by column ownership into two plans -- own columns (read the rank's
 * own previous y: no communication) and other columns (read the gather buffer, y += ) -- so the own-column product of
 * iteration t+1 runs while the all-gather of iteration t is still in flight on a private communication stream.
 * librccl.so.1 is opened with dlopen on first use (the copy already mapped in the process, e.g. PyTorch's, else the
 * system one; DASP_RCCL_LIB overrides): libdasp_amd.so itself does not depend on it. */
typedef struct dasp_mg_plan dasp_mg_plan_t;
enum { DASP_MG_ID_BYTES = 128 };   /* sizeof(ncclUniqueId) */

typedef struct dasp_mg_info {
    int precision, n_gpus, rank, rowA, colA;
    int row_begin, row_end;        /* this rank's rows */
    int stride;                    /* padded slice length in elements (multiple of 64) */
    long long nnz_own, nnz_other;  /* nonzeros in the rank's own column range / elsewhere (nnz_other = 0 without the split) */
    int overlap, has_comm, square;
    int stream_memops;             /* (after dasp_mg_upload; two-launch form) 1: the two streams hand over through hipStreamWriteValue64 /
                                      hipStreamWaitValue64 on two words of signal memory (DASP_MG_SYNC=memops; a Beta API); 0: through events */
    int fused_step;                /* (after dasp_mg_upload) 1: dasp_mg_spmv / dasp_mg_product run the one-launch step (below) */
    int exchange;                  /* 0: RCCL all-gather (or the test hook), 1: direct stores into the peers' gather buffers (dasp_mg_push_connect) */
} dasp_mg_info_t;

/* contiguous row ranges with equal nonzero counts: bounds[0]=0 <= ... <= bounds[n_parts]=rowA */
int dasp_partition_rows(int rowA, const int *csrRowPtr, int n_parts, int *bounds);

/* rank 0: ncclGetUniqueId into id[DASP_MG_ID_BYTES]; hand the bytes to every rank out of band (MPI_Bcast, a file, a
 * torch.distributed broadcast ...), then every rank calls dasp_mg_comm_init with them */
int dasp_mg_unique_id(void *id);

/* host part (no GPU needed): classifier + packers of this rank's slice.  csrRowPtr [rows+1] is local (starts at 0),
 * csrColIdx holds GLOBAL column ids, csrVal as dasp_plan_create.  opt: NULL = defaults (y_order and the column partition
 * fields are set by this call).  overlap: 1 = own / other column split (square matrices, n_gpus > 1), 0 = one plan. */
int dasp_mg_plan_create(dasp_mg_plan_t **mg, int precision, int rowA, int colA, int n_gpus, int rank, const int *row_bounds,
                        const int *csrRowPtr, const int *csrColIdx, const void *csrVal, const dasp_options_t *opt, int overlap);
void dasp_mg_destroy(dasp_mg_plan_t *mg);
/* the plans + the slice / gather buffers + the communication stream, on the CURRENT device */
int dasp_mg_upload(dasp_mg_plan_t *mg);
/* ncclCommInitRank(n_gpus, id, rank) on the current device: collective, every rank must call it */
int dasp_mg_comm_init(dasp_mg_plan_t *mg, const void *id);
/* x_0: colA host values -> the device layout the products read (synchronises the device) */
int dasp_mg_set_x(dasp_mg_plan_t *mg, const void *x_host);
/* one iteration, asynchronous: own-column product | wait for the previous all-gather | other-column product (y +=) on
 * `stream`, then ncclAllGather(y slice -> gather buffer) on the communication stream behind them.  Square matrices: the
 * gathered y is the next call's x.  Rectangular: x stays what dasp_mg_set_x stored.
 * ONE dasp_mg_plan is driven from ONE stream: pass the same `stream` to every dasp_mg_spmv / _product / _wait / _allgather of a plan.
 * Fused step (f64, square, column split, plans with 16-bit ids and without x windows / column panels / multi-piece long rows;
 * DASP_MG_FUSED=0 turns it off): the two products are ONE launch -- own-column workgroups first (y written through), then a
 * bounded set of persistent workgroups that wait inside the kernel for the count of finished own-column workgroups and for the
 * previous exchange's flag and run the other-column plan (y +=; the arithmetic of the two-launch form, bit-identical) -- and the
 * launch's last workgroup publishes "y ready" to a one-lane kernel at the head of the communication stream: `stream` carries
 * back-to-back kernels only.  In-kernel waits give up after 1 s (DASP_MG_TIMEOUT_MS) and set a sticky error instead of
 * hanging: see dasp_mg_check. */
int dasp_mg_spmv(dasp_mg_plan_t *mg, void *stream);
/* synchronises the device and reports whether a wait of the fused step timed out since the last check: DASP_OK, or DASP_ERR_STATE
 * after switching the plan to the two-launch form (results since the time-out are invalid: dasp_mg_set_x and start again). */
int dasp_mg_check(dasp_mg_plan_t *mg);
/* choose the form explicitly (1 needs a plan that qualifies); synchronises the device, the current y slice stays valid */
int dasp_mg_set_fused(dasp_mg_plan_t *mg, int on);
/* TEST HOOK, never needed with RCCL: lets dasp_mg_spmv run at n_gpus > 1 without a communicator.  The exchange becomes a copy of
 * the rank's slice into its own gather buffer and into `peer_gathered[0..n_peers)` (dasp_mg_gathered of other ranks' plans living on
 * the same device; NULL entries are skipped), followed by a kernel that holds the communication stream for `micros` us.  There is no
 * cross-rank synchronisation: the caller orders the ranks' steps itself. */
int dasp_mg_set_fake_exchange(dasp_mg_plan_t *mg, int micros, int n_peers, void *const *peer_gathered);
/* ---- direct exchange (r3): instead of an RCCL collective every rank STORES its slice into every rank's gather buffer through
 * peer-mapped pointers (hipIpc; xGMI: one hop, all links at once) and then a sequence number into the receiver's arrived[sender] word;
 * a one-wave kernel on the receiver waits for all senders.  Why: RCCL's kernels on gfx950 need 261-280 registers per lane and do not
 * start beside a product kernel that keeps every SIMD full -- the exchange then runs AFTER the product instead of under it
 * (DESIGN_MULTIGPU.md 5.3); these kernels hold <= 32.  Single node.  Usage, after dasp_mg_upload on every rank:
 *     dasp_mg_push_export(mg, blob)                    this rank's DASP_MG_IPC_BYTES
 *     ... all-gather the blobs among the ranks by any means (MPI_Allgather, torch.distributed.all_gather, files) ...
 *     dasp_mg_push_connect(mg, blobs)                  [n_gpus][DASP_MG_IPC_BYTES] in rank order; maps the peers and switches the plan over
 *     dasp_mg_set_x(mg, x)                             with the direct exchange a COLLECTIVE point: every rank calls it (a rank's next
 *                                                      exchange waits, on the host, until every peer has passed the same call;
 *                                                      DASP_MG_BARRIER_TIMEOUT_S, default 120)
 * Peers inside one process (one process driving several plans) are used through their plain pointers (on different devices: the caller
 * enables peer access between them).  A sender that does not deliver
 * within the time-out sets the sticky error: dasp_mg_check then returns DASP_ERR_STATE (the exchange is not switched by it: that is a
 * collective decision -- dasp_mg_set_exchange(mg, 0) on every rank, then dasp_mg_set_x).  Starting again with dasp_mg_set_x on every
 * rank is always possible: the flags carry (number of that collective call, exchanges since it), so ranks that stopped at different
 * steps agree again.
 * dasp_mg_set_exchange: 0 = RCCL, 1 = direct (after a connect); synchronises the device, the current x stays valid.  Collective like
 * dasp_mg_set_x: every rank switches at the same point of its call sequence. */
enum { DASP_MG_IPC_BYTES = 256 };
int dasp_mg_push_export(dasp_mg_plan_t *mg, void *blob);
int dasp_mg_push_connect(dasp_mg_plan_t *mg, const void *blobs);
int dasp_mg_set_exchange(dasp_mg_plan_t *mg, int mode);
/* For RCCL as the exchange: a compute stream of the plan that keeps `reserve_cus` CUs (rounded up to whole groups of 32: one CU per shader
 * engine per XCD) free for RCCL's kernels (hipExtStreamCreateWithCUMask), or NULL.  Pass it as `stream` to dasp_mg_spmv & co.  Without it
 * RCCL's kernel waits for the product kernel to drain (and, in the fused step, for the waiting workgroups to time out: use the two-launch
 * form then); with it the product runs on the remaining CUs (HV15R, 224 of 256: +8 %, tools/cumask_product.py).  Owned by the plan.
 * dasp_mg_spmv / dasp_mg_product with an RCCL communicator of several ranks run the fused step ONLY on this stream (two launches on any
 * other); the direct exchange has no such restriction. */
void *dasp_mg_reserved_stream(dasp_mg_plan_t *mg, int reserve_cus);
/* TEST HOOK (timing on a one-GPU box): the direct exchange with scratch memory of this rank standing in for every peer */
int dasp_mg_push_loopback(dasp_mg_plan_t *mg);
/* the products of one iteration only (no exchange): for callers that move dasp_mg_y_local into every rank's
 * dasp_mg_gathered themselves (tests; transports other than RCCL).  EXCEPT in the one-stream step (overlap = 2 with the direct exchange):
 * there the launch itself sends the previous slice to the peers and waits for theirs, so the call is dasp_mg_spmv under another name -- it
 * waits for the peers' epoch (dasp_mg_set_x) and refuses after a timed-out wait exactly as dasp_mg_spmv does. */
int dasp_mg_product(dasp_mg_plan_t *mg, void *stream);
/* the exchange alone, on `stream` (no product): the current y slice -> every rank's gather buffer; collective.  For timing the
 * all-gather by itself next to the products. */
int dasp_mg_allgather(dasp_mg_plan_t *mg, void *stream);
/* make `stream` wait for the all-gather still in flight (after it, dasp_mg_gathered holds the full y) */
int dasp_mg_wait(dasp_mg_plan_t *mg, void *stream);
/* the gathered y without the padding, rowA host values (synchronises the device) */
int dasp_mg_get_y(dasp_mg_plan_t *mg, void *y_host);
/* this rank's y slice without the padding, row_end - row_begin host values (synchronises the device) */
int dasp_mg_get_y_local(dasp_mg_plan_t *mg, void *y_host);
/* device pointers: this rank's current padded y slice [stride]; the gather buffer [n_gpus * stride]; what the products
 * read as x (the gather buffer for a square matrix, a colA vector otherwise) */
void *dasp_mg_y_local(dasp_mg_plan_t *mg);
void *dasp_mg_gathered(dasp_mg_plan_t *mg);
void *dasp_mg_x(dasp_mg_plan_t *mg);
/* BORROWED handle of the rank's plan over its own columns / the whole slice (which = 0) or over the other ranks' columns
 * (which = 1; NULL if there is none): stats, host arrays, timing */
dasp_plan_t *dasp_mg_subplan(dasp_mg_plan_t *mg, int which);
int dasp_mg_info(const dasp_mg_plan_t *mg, dasp_mg_info_t *out);

/* ---------------------------------------------------------------- device self-test
 * known-answer check of the MFMA operand / accumulator lane maps this library relies on
 * (v_mfma_f64_16x16x4_f64, v_mfma_f32_16x16x16_f16).  0 = maps as assumed. */
int dasp_selftest_mfma(void);

/* ---------------------------------------------------------------- synthetic inputs
 * Seeded stand-ins for the SuiteSparse matrices BASELINE.json names (no .mtx files and no
 * network on the build/bench machines).  Rows [row_begin,row_end) of the named matrix are
 * generated as CSR (global column ids, file-like order inside a row); every row can be
 * generated independently, so ranks of a multi-GPU run build only their slice.
 *   names: "cop20k_A" "nlpkkt160" "powerlaw_1M" "webbase-1M" "ljournal-2008" "HV15R" "Queen_4147" "rmat_2M", the
 *          round-1 worst-case variants "webbase-1M-uniform" "ljournal-2008-uniform", and "HV15R-unstructured" (HV15R's size and
 *          row lengths without the structured-grid numbering) -- dasp_synth_generator describes each
 *   scale: 1.0 = the collection's size; <1 shrinks the row count (tests). */
int dasp_synth_dims(const char *name, double scale, int *rows, int *cols);
/* one-line description of the generator behind `name` (seed, structure class, locality parameters); NULL for an unknown
 * name.  Thread-local storage, valid until the next call. */
const char *dasp_synth_generator(const char *name);
int dasp_synth_row_lengths(const char *name, double scale, int row_begin, int row_end, int *len_out);
int dasp_synth_rows(const char *name, double scale, int row_begin, int row_end,
                    const int *row_ptr_local /* [row_end-row_begin+1], exclusive scan of lengths */,
                    int *col_idx_out);

#ifdef __cplusplus
}
#endif
#endif /* DASP_AMD_H */

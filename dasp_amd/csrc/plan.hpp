// plan.hpp -- host-side DASP plan in the gfx950-native geometry.
//
// What the reference calls "preprocessing" (the host half of spmv_all,
// src/dasp_f64.h:499-1157 / src/dasp_f16.h:1029-1443) lives here: classify rows by
// length, sort the medium rows, split them into regular MFMA tiles + irregular tails,
// lay the long rows out for wave-sized pieces and the short rows as uniform-length slabs.
// The categories, the 1&3 pairing count and the output permutation (order_rid) are the
// reference's; the tile geometry is this build's own (DESIGN.md "Data layout in HBM").
#pragma once

#include <cstddef>
#include <cstdint>
#include <memory>
#include <string>
#include <utility>
#include <vector>

#include "../../include/dasp_amd.h"

namespace dasp {

// std::vector that does not zero-fill on resize(): the packed arrays are gigabytes, and a serial fill
// costs more than the (parallel) packing itself; every element is written by the packers.
template <class T>
struct NoInit : std::allocator<T> {
    template <class U> struct rebind { using other = NoInit<U>; };
    NoInit() = default;
    template <class U> NoInit(const NoInit<U> &) {}
    template <class U> void construct(U *p) noexcept { ::new (static_cast<void *>(p)) U; }
    template <class U, class... A> void construct(U *p, A &&...a) { ::new (static_cast<void *>(p)) U(std::forward<A>(a)...); }
};
template <class T> using raw_vector = std::vector<T, NoInit<T>>;

constexpr int kWave = 64;         // gfx950 wavefront
// packed bytes above which the streamed tiles use non-temporal loads: the 256 MiB Infinity Cache cannot keep the matrix between two
// SpMVs anyway (A/B around it: 212 MB plain 35.6 us vs nt 42.8; 257 MB plain 42.5 vs nt 51.9; 275 MB nt 54.9 vs plain 57.8)
constexpr long long kStreamBytes = 256ll << 20;
constexpr int kWavesPerWG = 4;    // 256-thread workgroups

// Paired chunks.  A medium block that is long enough for the kernel's software pipeline (nc regular chunks + nt tail steps > the
// one-shot limit) stores its first npair = nc rounded down to whole pipeline batches chunks (a one-shot f16 block: nc rounded down to
// even) in PAIRS -- [pair][lane][2 chunks][vpl],
// values and ids alike -- so that one 16-byte load per lane brings two chunks (the L1 takes a 16-byte load in as many passes as an
// 8-byte one: half the tag lookups per streamed byte, profiles/r02_pairs.md).  The remaining chunks, every chunk of a one-shot f64
// block and every chunk of an LDS-windowed plan stay lane-linear [chunk][lane][vpl].  kMedBatch / kMedShot are the kernel's Tr<T>::BATCH / SHOT (static_assert there).
// Shared by the host packer, the device packer and the plan validator; mirrored in tests/util.py.
#ifndef DASP_SHOT16
#define DASP_SHOT16 4
#endif
constexpr int kMedBatch64 = 4, kMedShot64 = 8, kMedBatch16 = 2, kMedShot16 = DASP_SHOT16;      // (other values measured: profiles/r02_pairs.md section 3)
#if defined(__HIPCC__)
__host__ __device__
#endif
// mode (Plan::pair_mode): 0 = nothing paired (LDS-windowed plans: latency-bound short blocks, their kernel is the unpaired one);
// 1 = pipelined blocks and one-shot f16 blocks; 2 = also the one-shot f64 blocks without tail steps -- plans far beyond the Infinity
// Cache (nlpkkt160 f64 0.4905 -> 0.4543 ms on a slow-population box, 0.4328 -> 0.4261 on a fast one; at 278 MB it costs 4-8 %)
inline int med_npair(int nc, int nt, int vbytes, int mode)
{
    if (mode == 0) return 0;
    const int batch = vbytes == 8 ? kMedBatch64 : kMedBatch16, shot = vbytes == 8 ? kMedShot64 : kMedShot16;
    // one-shot blocks: f16 pairs them (nlpkkt160 f16 0.2355 -> 0.2177 ms); f64 only as a whole (no tail steps: one test per block) and only in mode 2
    return nc + nt > shot ? nc / batch * batch : (vbytes == 2 || (mode == 2 && nt == 0) ? nc & ~1 : 0);
}
// where the one-byte id of lane `lane` of the narrow chunk at position q (< n8) sits inside its block's region of med_cid8
#if defined(__HIPCC__)
__host__ __device__
#endif
// (a pipelined block: its batch of four chunks interleaved per lane, one dword per lane and batch; a ONE-SHOT block (r4): its pair of
// chunks interleaved per lane, one 16-bit word per lane and pair)
inline size_t med_cid8_index(int q, int lane, int ch, bool oneshot = false)
{
    return oneshot ? (size_t)(q & ~1) * ch + 2 * (size_t)lane + (q & 1) : (size_t)(q & ~3) * ch + 4 * (size_t)lane + (q & 3);
}
// is a block of nc regular chunks and nt tail steps issued in one shot by the f64 kernel (no software pipeline)?
#if defined(__HIPCC__)
__host__ __device__
#endif
inline bool med_oneshot64(int nc, int nt) { return nc + nt <= kMedShot64; }
// how many narrow chunks a block keeps: those of its paired region -- in whole batches of four in a pipelined block, in whole pairs in a
// one-shot block (r4: nlpkkt160's rows of 5-28 nonzeros are one-shot blocks, 68 % of their chunks span <= 254 columns; a one-shot f64 block
// has a paired region only in pair_mode 2, i.e. in plans far beyond the Infinity Cache)
#if defined(__HIPCC__)
__host__ __device__
#endif
inline int med_n8(int narrow_in_paired_region, int nc, int nt) { return nc + nt > kMedShot64 ? narrow_in_paired_region / kMedBatch64 * kMedBatch64 : narrow_in_paired_region / 2 * 2; }
// where element j (< vpl) of lane `lane` of regular chunk c sits inside its block's region of med_val / med_cid / med_cid16 (in elements)
#if defined(__HIPCC__)
__host__ __device__
#endif
inline size_t med_elem_index(int npair, int c, int lane, int j, int vpl, int ch)
{
    return c < npair ? (size_t)(c & ~1) * ch + (size_t)vpl * (2 * lane + (c & 1)) + j : (size_t)c * ch + (size_t)vpl * lane + j;
}
#ifndef DASP_SHORT_TPW
#define DASP_SHORT_TPW 4
#endif
constexpr int kShortTpw = DASP_SHORT_TPW;      // wave-segmented short tiles (64 elements each) handled by one wave (device.hpp DevArgs::short_tpw)
constexpr int kMedRows = 16;      // rows of one MFMA tile (v_mfma_*_16x16x*)
constexpr int kLongAlign = 4;     // long rows start on a multiple of 4 elements
// ... and only pieces of at least four whole chunks take them: a row of 300 in f16 (one chunk and a tail) is two steps whose id unpacking costs more than its 600 bytes
// save (rows of 300 f16 131 -> 150 us with 16-bit ids; rows of 2000 95.6 -> 88.7, 400 rows of 200 000 92.9 -> 83.9; f64 rows of 300, four chunks and a tail: 196 / 198)
constexpr int kLong16MinChunks = 4;
constexpr unsigned short kLongPad16 = 0xFFFFu;      // a pad among the 16-bit ids of a narrow long piece (Plan::long_cid16)
constexpr int kSlabMaxLen = 32;     // longest medium row that can be stored as a uniform-length slab (opt.slab_max_len)
// slab groups: row lengths 1,2,3,4 and 0 (zero rows only get y=0) -- the reference's short rows -- then 5..kSlabMaxLen
constexpr int kNumShortGroups = 5 + (kSlabMaxLen - 4);

struct Geometry {
    int vbytes;      // 8 / 2
    int med_k;       // K extent of one MFMA tile: 4 (f64 16x16x4) / 16 (f16 16x16x16)
    int chunk;       // kMedRows * med_k elements = one MFMA = one coalesced wave load
    int short_rows;  // rows one wave handles in a short slab: 128 (2 per lane) / 256 (4 per lane)
};
inline Geometry geometry_for(int precision)
{
    return precision == 64 ? Geometry{8, 4, 64, 128} : Geometry{2, 16, 256, 256};
}

// slot(t) for the t-th row of a short group: two pieces, each linear or "paired"
// (the reference interleaves 8 (f64) / 32 (f16) len-1 rows with as many len-3 rows).
struct SlotMap {
    int split;                 // t <  split -> piece 0 with u = t ; else piece 1 with u = t - split
    int base[2], grp[2], off[2];   // grp == 0: base + u ; else base + (u / grp) * 2 * grp + off + u % grp
    int slot(int t) const
    {
        const int p = t < split ? 0 : 1;
        const int u = p ? t - split : t;
        return grp[p] ? base[p] + (u / grp[p]) * 2 * grp[p] + off[p] + u % grp[p] : base[p] + u;
    }
};

struct ShortGroup {
    int len = 0;            // nonzeros per row (0..4)
    int count = 0;          // rows
    int tiles = 0;          // ceil(count / rpt)
    int tile0 = 0;          // index of this group's first tile among all short tiles
    long long elem_off = 0; // first element in short_val / short_cid
    SlotMap map{};
    int seg = 0;            // 1: wave-segmented layout (short_seg_*): one nonzero per lane, rows back to back, DPP row-shift sums
    int rpt = 0;            // rows one wave handles: geo.short_rows for a slab, short_seg_rows(len) for a segmented group
};

// Wave-segmented short rows (opt.short_seg; the north_star's "wavefront-segmented dot product with DPP reductions", reference branches
// dasp_f64.h:281-483).  A tile is 64 ELEMENTS = one per lane of one wave: every 16-lane DPP row holds 16 / L whole rows of L nonzeros back to
// back (L = 3: five rows and one idle lane), so that a row is summed by two DPP row_shl steps inside its 16 lanes and the lanes that hold a
// row's first nonzero store y.  Element (row t of the group, entry k) -> index inside the group's region; shared by the host packer, the
// device packer, the plan validator and the test decoder (tests/util.py).
#if defined(__HIPCC__)
__host__ __device__
#endif
inline int short_seg_rows(int L) { return 4 * (16 / L); }                        // rows per tile: 64, 32, 20, 16
#if defined(__HIPCC__)
__host__ __device__
#endif
inline size_t short_elem_index(bool seg, int L, int SR, long long t, int k)
{
    if (!seg) return ((size_t)(t / SR) * L + k) * SR + (size_t)(t % SR);                     // slab: [tile][k][SR]
    const int per16 = 16 / L, rpw = 4 * per16, r = (int)(t % rpw);
    return (size_t)(t / rpw) * kWave + (size_t)((r / per16) * 16 + (r % per16) * L + k);  // [tile][lane]
}
// elements of one tile of a group
#if defined(__HIPCC__)
__host__ __device__
#endif
inline int short_tile_elems(bool seg, int L, int SR) { return seg ? kWave : L * SR; }

constexpr int kRowTile = 64;        // output positions per row tile: one wave, one lane per position
constexpr int kRowTileAuto64 = 8;    // opt.row_tile_max = 0, f64: 16 KB of LDS per workgroup -- at 16 (32 KB) a CU holds 5 instead of 6 workgroups of the f64 kernel, which costs an HBM-bound panel 7 % (HV15R-unstructured in two forced panels 0.554 -> 0.592 ms) and gains powerlaw_1M 0.6 %
constexpr int kRowTileAuto = 16;    // opt.row_tile_max = 0, f16 (tools/row_tile_ab.py: ljournal-2008 f16 0.483 / 0.469 / 0.460 / 0.452 / 0.451 / 0.451 / 0.496 ms at 6 / 8 / 12 / 16 / 20 / 24 / 32, 0.507-0.511 without)
constexpr int kRowTileMax = 32;     // longest row a tile may take (LDS: 4 waves x 64 x bound products per workgroup)

// ---- two-phase (gather-free) form of a plan: opt.two_phase, f16 (no reference counterpart).
// Every kernel that gathers x per nonzero pays ~0.8 L1 misses per nonzero on graph matrices whose rows scatter over all of x; here no gather
// leaves the CU.  The nonzeros are cut into TILES (row block r, column block c): row blocks are ranges of at most rb_max consecutive OUTPUT
// positions (slots of order_rid, or rows in DASP_Y_NATURAL) holding about the same number of nonzeros, column blocks are ranges of `cb`
// columns.  A tile's nonzeros keep their CSR order (rows in output order) and are padded to whole SEGMENTS of kTpSeg elements.
//   phase 1 (dasp_tp_expand_kernel): one workgroup per UNIT (a run of <= kTpUnitSegs segments of one column block, CB-major order): stages
//       x[c * cb, (c + 1) * cb) in LDS and writes xs[dst[s] * kTpSeg + i] = x[c * cb + lcol[s * kTpSeg + i]]: the x value of every nonzero,
//       as a stream in RB-major order.
//   phase 2 (dasp_tp_reduce_kernel): one workgroup per row block: y slice in LDS as f64, streams (val, lrow, xs) -- all contiguous -- and
//       adds val * xs to position lrow with LDS f64 atomics (ds_add_f64: ~4x the rate of ds_add_f32 on gfx950), then stores y once.
// Pads: val 0, lrow kTpPadRow (skipped), lcol 0.
#ifndef DASP_TP_SEG
#define DASP_TP_SEG 64
#endif
constexpr int kTpSeg = DASP_TP_SEG;   // elements per segment (64: 128 bytes of f16 / u16; 32 is the other supported size): the unit a tile is padded to and phase 1 places
constexpr int kTpUnitSegs = 65536 / kTpSeg;     // segments per phase-1 workgroup (64 Ki elements against the 64-KiB x slice it stages)
constexpr int kTpColBlock = 32768;    // default cb: 64 KiB of LDS, two phase-1 workgroups per CU
constexpr int kTpRowBlock = 4096;     // default rb_max: 32 KiB of f64 accumulators, five phase-2 workgroups per CU
constexpr unsigned short kTpPadRow = 0xFFFFu;
struct TwoPhase {
    int cb = 0, rb_max = 0;
    std::vector<int> rb_row0;         // [n_rb + 1] first output position of every row block
    std::vector<int> rb_seg0;         // [n_rb + 1] first segment (RB-major) of every row block
    std::vector<int> unit;            // [n_units * 3] column block, first segment, end segment (CB-major)
    std::vector<int> dst;             // [segments] CB-major segment -> RB-major segment
    raw_vector<uint16_t> lcol;        // [segments * kTpSeg] CB-major: column - c * cb
    raw_vector<uint16_t> lrow;        // [segments * kTpSeg] RB-major: output position - rb_row0[r]
    raw_vector<char> val;             // [segments * kTpSeg * vbytes] RB-major
    size_t segments = 0;
    int n_rb() const { return rb_row0.empty() ? 0 : (int)rb_row0.size() - 1; }
    int n_units() const { return (int)(unit.size() / 3); }
};

// ---- column-blocked long rows of a column-panel plan (opt.long_cb, r5; the north_star's LDS-staged x gathers applied to the long rows; no reference
// counterpart: the reference's long rows gather x from global memory, dasp_f64.h:90-144).  A hub row's nonzeros scatter over all of x; in a column panel
// its gathers still miss the L1 once per nonzero.  Here the rows of >= lcb.h nonzeros leave the panels: their nonzeros are cut by COLUMN BLOCK of `cb`
// columns into PIECES -- piece (c, i) = the entries of long row i inside block c, CSR order, padded to whole STEPS of kLcbStep elements -- stored CB-major.  One
// workgroup per UNIT (a run of pieces of one column block holding ~kLcbUnitElems elements) stages the block's slice of x in LDS and streams the unit's steps --
// every step an independent 16-byte-per-lane load of values and local columns, value x LDS-x, a wave (f64) / quarter-wave (f16) sum parked in LDS --
// then adds every piece's step sums in order (no atomics: the result is the same in every run) and writes partial[c * n_rows + i]; a second kernel adds a row's n_cb partials and stores the result into the row's slot
// of panel 0's partial buffer (which no panel writes: the row is empty there), so dasp_panel_sum_kernel folds it into y like any other partial result.
// Streamed per nonzero: value + u16 local column (10 B in f64 against the 12 of B_alg), nothing gathers from global memory.
#ifndef DASP_LCB_UNIT
#define DASP_LCB_UNIT 32768
#endif
constexpr int kLcbUnitElems = DASP_LCB_UNIT;      // elements per phase workgroup (A/B: 65536 -- fewer slices of x, a longer tail)
constexpr int kLcbStep = 128;             // elements per STEP: a piece is padded to whole steps, so that a step belongs to one piece (one wave of f64 / a quarter wave of f16 per step)
constexpr int kLcbUnitPieces = 1024;      // most pieces per unit (the sums of a unit's steps live in LDS beside the slice of x: <= kLcbUnitElems / kLcbStep + kLcbUnitPieces of them)
constexpr unsigned short kLcbPadCol = 0xFFFFu;
struct LongCB {
    int cb = 0, n_cb = 0, h = 0;      // columns per block, blocks, the shortest row taken
    std::vector<int> row_dst;         // [n_rows] output position (the parent's slot, or the row in DASP_Y_NATURAL) of every long row
    std::vector<int> row_id;          // [n_rows] the row itself (host only: decoding, validation)
    std::vector<int> ptr;             // [n_cb * n_rows + 1] element offsets of the pieces, CB-major
    std::vector<int> unit;            // [n_units * 3] column block, first piece, end piece
    raw_vector<uint16_t> lcol;        // [elems] column - c * cb (pads: kLcbPadCol)
    raw_vector<char> val;             // [elems * vbytes]
    size_t elems = 0;
    int n_rows() const { return (int)row_dst.size(); }
    int n_units() const { return (int)(unit.size() / 3); }
};

struct DevicePlan;  // kernels.hip
}  // namespace dasp
struct dasp_plan;      // the C handle (defined at the end of this file): one Plan
namespace dasp {

struct Plan {
    int precision = 64;
    Geometry geo{};
    int m = 0, n = 0, nnz = 0;
    dasp_options_t opt{};
    std::vector<int> part_bounds;   // copy of opt.part_bounds
    dasp_stats_t stats{};

    std::vector<int> order;         // [m]   order_rid

    // long rows: CSR-ordered, each row padded to kLongAlign (val 0, cid -1); pieces of <= long_piece
    raw_vector<char> long_val;      // vbytes per element
    raw_vector<int> long_cid;
    // 16-bit ids of the long pieces (r6): chunk i of piece p -- elements [piece_ptr[p] + i CH, + CH) -- has the base column long_base[piece_c16[2 p] + i]
    // (its smallest column; 0 for a chunk of pads only); a piece all of whose chunks span <= 65534 columns is NARROW (piece_c16[2 p + 1] = 1) and its ids are also stored as
    // u16 offsets from the chunk's base in long_cid16[element] (0xFFFF = pad): the kernel streams 10 / 4 instead of 12 / 6 bytes per nonzero.  Lane assignment and the order of
    // a row's additions are those of the 32-bit form, which stays complete in long_cid (wide pieces, the multi-GPU step kernels, decoders); a wide piece's u16 ids are 0.
    raw_vector<uint16_t> long_cid16;  // [cnt_long]
    std::vector<int> long_base;       // [chunks of all pieces]
    std::vector<int> piece_c16;       // [2 P] {first chunk, narrow}
    std::vector<int> piece_ptr;     // [P+1] element offsets
    std::vector<int> piece_dst;     // [P]   >= 0: y index ; < 0: partial sum ~dst
    std::vector<int> multi_ptr;     // [R2+1] ranges of partial sums of rows cut into several pieces
    std::vector<int> multi_dst;     // [R2]  y index

    // medium rows: sorted by length (desc, stable); blocks of 16 rows
    std::vector<int> med_ptr;       // [nb+1] in chunks
    raw_vector<char> med_val;       // chunk-major, lane-linear inside a chunk
    raw_vector<int> med_cid;        // 32-bit ids (cid16 off)
    bool cid16 = false;
    raw_vector<uint16_t> med_cid16;  // u16 offsets from med_base[chunk], 0xFFFF = pad (cid16 on)
    std::vector<int> med_base;        // [chunks]
    // one-byte ids (f64, cid16 on, pipelined paired chunks; r2): a chunk of the paired region whose columns span <= 254 stores its offsets in
    // ONE byte (0xFF = pad), plane med_cid8, a pipeline batch of four chunks interleaved per lane: [batch][lane][4 chunks], one dword per
    // lane.  A block's chunks are independent MFMA steps, so the packer moves n8 of them -- whole batches -- to the front of the paired
    // region: position q of the block holds the block's chunk med_korig[q] (values, base and ids move together), positions [0, n8) are
    // narrow.  med_c8ptr[b] = narrow chunks before block b; the block's wide ids start at (med_ptr[b] - med_c8ptr[b]) * CH in med_cid16.
    raw_vector<uint8_t> med_cid8;
    std::vector<int> med_c8ptr;       // [nb+1] (cid16 plans)
    std::vector<int> med_korig;       // [chunks] (cid16 plans)
    size_t cnt_reg8 = 0;              // elements of the narrow chunks (cnt_reg counts all regular elements)
    int n_mfma_rows = 0;            // medium rows handled as MFMA blocks (the shortest are slabs, see grp[5..]; the longest may be pieces)
    int med_slot0 = 0;              // slot of the first MFMA medium row: row_long + the medium rows stored as pieces (opt.piece_min_len)
    std::vector<int> irr_ptr;       // [n_mfma_rows+1]
    raw_vector<char> irr_val;
    raw_vector<int> irr_cid;
    // windowed mode (LDS-staged x): medium positions follow the windowed order; med_dst[pos] = y index,
    // win_cmin/win_len = the x span of window w (len 0: span too wide, that window gathers from global memory)
    bool windowed = false;
    int pair_mode = 0;              // which medium blocks store their chunks in pairs (med_npair): decided before packing
    bool win_hybrid = false;        // windows stage their densest span only: gathers outside it read global memory
    // 16-bit ids of an LDS-staged window are offsets from the WINDOW's first staged column (med_base[chunk] = win_cmin[window] for all its
    // chunks), i.e. they index the staged span directly: the kernel needs neither the per-chunk base load nor the subtraction
    bool win_rel16 = false;
    int row_window = 0, lds_bytes = 0;
    std::vector<int> med_dst, win_cmin, win_len;

    // short rows: per length one slab, tile-major [tile][k][short_rows]
    ShortGroup grp[kNumShortGroups];
    raw_vector<char> short_val;
    raw_vector<int> short_cid;

    // element counts of the nnz-sized arrays (the host vectors are empty when the plan was packed on the device)
    size_t cnt_long = 0, cnt_reg = 0, cnt_irr = 0, cnt_short = 0;
    size_t cnt_long_chunks = 0;      // chunks of all long pieces (entries of long_base)

    // column panels (opt.col_panels): this plan then holds only order / stats of the whole matrix and owns one natural-order
    // plan per column range [panel_bounds[k], panel_bounds[k+1]); panel k writes its partial result for row r to
    // ypart[k][dst_map[r]] (dst_map = the parent's slot of each row; empty when the parent itself is in natural order)
    std::vector<std::unique_ptr<dasp_plan>> panels;
    std::vector<int> panel_bounds;     // [panels.size()+1] pairs flattened: begin/end per kept panel (empty panels are dropped)
    bool panel = false;                // this plan is one column panel of another
    // (building a panel only) the order in which the classifier walks the rows: the parent's slot order instead of the row order, so that rows
    // of equal length in THIS panel follow each other in the parent's slots -- a workgroup's partial results then fall into a narrow range of
    // the parent-ordered partial buffer instead of all over it (the panels' 2-byte stores were a quarter of ljournal-2008's time)
    const int *scan_order = nullptr;
    std::vector<int> dst_map;          // [m] set on a panel: row -> y index (instead of the row id) in natural order

    // row tiles (a column panel only, opt.row_tile_max): the panel's rows of <= rt_max nonzeros, in the PARENT's output order (position j =
    // the parent's slot j, or row j when the parent writes in row order), tiles of kRowTile positions.  rt_ptr[t]: first element of tile t;
    // rt_start[j]: first element of position j relative to its tile; rt_mask[t] bit i: position 64 t + i is stored by the tile (it belongs to a
    // row of <= rt_max nonzeros, empty rows included) -- the other positions belong to this panel's blocks / pieces / slabs.  Elements in CSR order.
    int rt_max = 0;
    std::vector<int> rt_ptr;           // [tiles + 1]
    std::vector<uint16_t> rt_start;    // [tiles * 64]
    std::vector<uint64_t> rt_mask;     // [tiles]
    raw_vector<char> rt_val;           // vbytes per element
    raw_vector<int> rt_cid;
    size_t cnt_rt = 0;

    // column-blocked long rows of a column-panel parent (LongCB above; empty otherwise)
    LongCB lcb;

    // two-phase form (TwoPhase above): the plan then holds order / stats of the whole matrix and the tile streams, nothing else
    bool two_phase = false;
    TwoPhase tp;

    bool host_dropped = false;
    DevicePlan *dev = nullptr;

    ~Plan();
};

// ---- device-side packing (dasp_plan_create_device): the CSR stays on the GPU; the host keeps doing the O(rows) decisions
// from the row pointer alone and hands the O(nnz) work to the kernels in devpack.hip through these hooks.
struct DevCsr { const int *rp, *ci; const void *val; };      // device pointers
struct PackMeta {                                             // what the device packers need, in packing order
    const std::vector<int> *ridL = nullptr; const std::vector<long long> *startL = nullptr;
    const raw_vector<int> *ridM = nullptr, *lenM = nullptr;      // (not zero-filled on construction: 33 MB each for 8 M rows)
    const std::vector<int> *glist[kNumShortGroups] = {};
};
int devpack_validate(const Plan &p, const DevCsr &d);
int devpack_window_spans(const Plan &p, const DevCsr &d, const raw_vector<int> &ridW, int R, int *lo, int *hi, long long *wnnz);
// over the sampled rows: nonzeros, and how many of them start a new 128-byte line of x relative to their predecessor in the row
int devpack_line_scatter(const Plan &p, const DevCsr &d, const std::vector<int> &rows, long long *lines, long long *entries);
// over pairs of equally long rows (rows[2i], rows[2i+1]): entries compared, and how many lie within 16 columns of the other row's
// entry at the same position
int devpack_row_coherence(const Plan &p, const DevCsr &d, const std::vector<int> &rows, int within, long long *near, long long *entries);
int devpack_chunk_spans(const Plan &p, const DevCsr &d, const raw_vector<int> &ridM, const raw_vector<int> &lenM,
                        const std::vector<int> &nchunks, int *k16, unsigned long long *narrow_mask);      // narrow_mask: nullptr or [blocks] (plan.cpp)
int devpack_all(Plan &p, const DevCsr &d, const PackMeta &m);
int devpack_finish_panels(Plan &p);
int devpack_fetch_csr(const Plan &p, const DevCsr &d, int *ci, void *val);      // column ids and values of a device CSR -> host arrays of nnz elements
// the calling thread's HIP device / make `device` the calling thread's (panel workers of a device-built plan)
int devpack_current_device();
void devpack_use_device(int device);
// remapped column ids at the nonzero positions idx[] (or start + i * stride for i < count when idx is null), copied to the host
int devpack_gather_columns(const Plan &p, const DevCsr &d, const std::vector<long long> *idx, long long start, long long stride, long long count, std::vector<int> &out);      // uploads the parent of device-built panels (its partial-result buffers)
// column-panel split of a device CSR: P sub-matrices by column range (remapped columns, row order kept); `keep` owns the device arrays
int devpack_panel_split(const Plan &p, const DevCsr &d, const std::vector<int> &bnd, int P, std::vector<std::vector<int>> &rpP_host,
                        std::vector<DevCsr> &out, std::vector<std::shared_ptr<void>> &keep);

// row tiles of a device-built column panel (plan.cpp build_panels): `panel` = the split's sub-matrix, rp_rest = the row pointer of the rows that
// stay with the panel's plan (host), at[row] = first element of a tiled row in the tiles' arrays (-1: not tiled), cnt = their elements.
// On return `panel` is the sub-matrix of the remaining rows only and `out` holds the tiles' elements (device, owned by `keep`).
struct DevRowTiles { void *val = nullptr; int *cid = nullptr; };
int devpack_row_tiles(const Plan &p, DevCsr &panel, const std::vector<int> &rp_rest, const std::vector<int> &at, size_t cnt,
                      std::vector<std::shared_ptr<void>> &keep, DevRowTiles *out);
// ... and their copy into the uploaded panel plan's arena (ArenaMap::rt_val / rt_cid)
int devpack_place_row_tiles(Plan &q, const DevRowTiles &src);

// opt.sort_columns on a device CSR: when a row's columns do not ascend, `out` = the same matrix with every row's (column, value) pairs sorted by column (stable segmented
// sort; new device arrays owned by `keep`) and *did = 1; otherwise *did = 0
int devpack_sort_columns(const Plan &p, const DevCsr &d, std::vector<std::shared_ptr<void>> &keep, DevCsr *out, int *did);

// builds every host array of `p` from CSR.  T = double or _Float16.  With `dev` set, rp is a host copy of the row pointer,
// ci / val are ignored and the nnz-sized arrays are produced on the device (the plan comes back uploaded).
int build_plan(Plan &p, const int *rp, const int *ci, const void *val, const DevCsr *dev = nullptr);

// two-phase form (twophase.cpp): the automatic rule (1: use it), the packer (after build_impl's meta pass: p.order / p.stats are set) and the
// checks a loaded plan file must pass
int decide_two_phase(const Plan &p, const int *rp, int scattered);
// windowed plans: the workgroup's dynamic LDS = the window's copy of x (Plan::lds_bytes, a multiple of 256, <= kWinLdsMax); the kernels keep a word of static LDS beside it
// (the unit counter of the window's waves)
constexpr int kWinLdsMax = 160 * 1024 - 256;
// short tiles folded into the window workgroups as fillers (upload_plan, spmv_body): tiles per window, or 0 when the tiles are too many beside the windows and keep
// workgroups of their own
// Only while the windows are at most two per CU: there the tiles' own workgroups are the ones left over at the end (cop20k_A x1 9.3 against 10.8 us, x2 16.0 / 24.9, x4 26.8 /
// 29.0; f16 x4 18.2 / 26.6); with several rounds of window workgroups they fill the gaps by themselves and folding costs (x16, 1695 windows: 112.7 against 107.7 us).
inline int win_fold_tiles(int n_windows, int n_short_tiles)
{
    return n_windows > 0 && n_windows <= 512 && n_short_tiles > 0 && n_short_tiles <= 2 * n_windows ? (n_short_tiles + n_windows - 1) / n_windows : 0;
}
constexpr int kTpDeclined = 1;          // build_two_phase under the automatic rule: the padded streams would pass 3 x the nonzeros (or the tile table 64 M entries) -- not an error
int build_two_phase(Plan &p, const int *rp, const int *ci, const void *val, const unsigned char *skip = nullptr);
bool validate_two_phase(const Plan &p, std::string &why);

// column-blocked long rows (longcb.cpp): which rows (in_lcb[row] = 1) a column-panel plan of P panels hands to them (0 rows: none), the packer, the checks
int decide_long_cb(const Plan &p, const int *rp, int P, std::vector<unsigned char> &in_lcb, int share_den = 4, int per_block = 64);
int build_long_cb(Plan &p, const int *rp, const int *ci, const void *val, const std::vector<unsigned char> &in_lcb, const int *slot_of_row);      // DASP_OK, an error, or 1: not representable (p.lcb left empty)
bool validate_long_cb(const Plan &p, int n_panels, std::string &why);

// loader (mmio.cpp).  val_out: malloc'd array of double or binary16.
int load_mtx(const char *path, int precision, int *m, int *n, int *nnz, int *sym, int **rp, int **ci, void **val);

int save_csr_bin(const char *path, int precision, int m, int n, int nnz, int sym, const int *rp, const int *ci, const void *val);
int load_csr_bin(const char *path, int precision, int *m, int *n, int *nnz, int *sym, int **rp, int **ci, void **val);

int save_plan(Plan &p, const char *path);
int load_plan(Plan &p, const char *path);

void set_error(const std::string &s);

// threads helper
int resolve_threads(int requested);

}  // namespace dasp

struct dasp_plan {
    dasp::Plan impl;
};

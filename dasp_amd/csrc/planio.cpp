// planio.cpp -- serialised plans (the reference never stores its packed format).
// A plan file holds every host array of a Plan, so a later run (or every rank of a multi-GPU job)
// skips classification and packing: load -> upload -> spmv.
// layout: "DASPPLN7" | int32 sizeof(dasp_stats_t), kNumShortGroups, sizeof(ShortGroup) | plan, where plan = int32 precision, m, n, nnz,
//         y_order, windowed, row_window, lds_bytes, cid16, n_parts, part_stride, stream_policy, n_panels, n_mfma_rows, win_hybrid, med_slot0, pair_mode, win_rel16, rt_max, two_phase, tp.cb, tp.rb_max, lcb.cb, lcb.n_cb, lcb.h | dasp_stats_t |
//         ShortGroup[kNumShortGroups] | for each array, in a fixed order: int64 byte count + bytes | the n_panels column panels, each
//         a nested plan.
// A file is not trusted more than a caller's CSR: after reading, every count, pointer array and column id the kernels index
// with is re-derived and checked (validate_plan), because upload_plan / the kernels would otherwise turn a stale or corrupt
// file into wild host, device or LDS accesses.
#include <algorithm>
#include <cstdio>
#include <cstring>
#include <string>

#include "plan.hpp"

namespace dasp {

namespace {
// bumped with EVERY change of the header, the array list or the element order inside an array (5: 18-int header, one-byte ids, chunk pairs; 6: 19-int header, row tiles; 7: 25-int header, the two-phase streams, the column-blocked long rows)
const char kPlanMagic[8] = {'D', 'A', 'S', 'P', 'P', 'L', 'N', '8'};      // v8 (r6): + long_cid16 / long_base / piece_c16

struct Writer {
    FILE *f; bool ok = true;
    void raw(const void *p, size_t n) { if (ok && n && std::fwrite(p, 1, n, f) != n) ok = false; }
    template <class V> void vec(const V &v) { long long b = (long long)(v.size() * sizeof(typename V::value_type)); raw(&b, 8); raw(v.data(), (size_t)b); }
};
struct Reader {
    FILE *f; long long left; bool ok = true;      // left: bytes of the file not yet consumed -- no array may claim more
    void raw(void *p, size_t n)
    {
        if (!ok || !n) return;
        if ((long long)n > left || std::fread(p, 1, n, f) != n) { ok = false; return; }
        left -= (long long)n;
    }
    template <class V> void vec(V &v)
    {
        long long b = -1; raw(&b, 8);
        if (!ok || b < 0 || b > left || b % (long long)sizeof(typename V::value_type)) { ok = false; return; }
        v.resize((size_t)b / sizeof(typename V::value_type));
        raw(v.data(), (size_t)b);
    }
};

template <class IO> void arrays(IO &io, Plan &p)
{
    io.vec(p.part_bounds); io.vec(p.order); io.vec(p.dst_map); io.vec(p.panel_bounds);
    io.vec(p.long_val); io.vec(p.long_cid); io.vec(p.long_cid16); io.vec(p.long_base); io.vec(p.piece_c16); io.vec(p.piece_ptr); io.vec(p.piece_dst); io.vec(p.multi_ptr); io.vec(p.multi_dst);
    io.vec(p.med_ptr); io.vec(p.med_val); io.vec(p.med_cid); io.vec(p.med_cid16); io.vec(p.med_cid8); io.vec(p.med_c8ptr); io.vec(p.med_korig); io.vec(p.med_base);
    io.vec(p.irr_ptr); io.vec(p.irr_val); io.vec(p.irr_cid);
    io.vec(p.med_dst); io.vec(p.win_cmin); io.vec(p.win_len);
    io.vec(p.short_val); io.vec(p.short_cid);
    io.vec(p.rt_ptr); io.vec(p.rt_start); io.vec(p.rt_mask); io.vec(p.rt_val); io.vec(p.rt_cid);
    io.vec(p.tp.rb_row0); io.vec(p.tp.rb_seg0); io.vec(p.tp.unit); io.vec(p.tp.dst); io.vec(p.tp.lcol); io.vec(p.tp.lrow); io.vec(p.tp.val);
    io.vec(p.lcb.row_dst); io.vec(p.lcb.row_id); io.vec(p.lcb.ptr); io.vec(p.lcb.unit); io.vec(p.lcb.lcol); io.vec(p.lcb.val);
}
}  // namespace

// Everything upload_plan and the kernels rely on, re-derived from the arrays themselves.  `why` names the first violation.
static bool validate_plan(const Plan &p, int n_panels, bool is_panel, std::string &why)
{
    auto fail = [&](const char *w) { why = w; return false; };
    const Geometry geo = p.geo;
    const long long vb = geo.vbytes, CH = geo.chunk, SR = geo.short_rows;
    const long long m = p.m;
    if (p.m < 0 || p.n < 0 || p.nnz < 0) return fail("negative dimension");
    if (p.opt.y_order != DASP_Y_PERMUTED && p.opt.y_order != DASP_Y_NATURAL) return fail("y_order");
    if (p.opt.n_parts < 0 || p.opt.n_parts > (1 << 20) || (p.opt.n_parts > 0 && p.opt.part_stride <= 0)) return fail("column partition");
    if (p.opt.n_parts > 0) {
        if (p.part_bounds.size() != (size_t)p.opt.n_parts + 1 || p.part_bounds.front() != 0 || p.part_bounds.back() != p.n) return fail("part_bounds size / span");
        for (int g = 0; g < p.opt.n_parts; ++g)
            if (p.part_bounds[g + 1] < p.part_bounds[g] || p.part_bounds[g + 1] - p.part_bounds[g] > p.opt.part_stride) return fail("part_bounds not monotone or wider than part_stride");
    } else if (!p.part_bounds.empty()) return fail("part_bounds without n_parts");
    const long long xlen = p.opt.n_parts > 0 ? (long long)p.opt.n_parts * p.opt.part_stride : (long long)p.n;
    if (xlen >= (1ll << 31)) return fail("x length");
    if (p.order.size() != (size_t)m) return fail("order size");
    {
        std::vector<bool> seen((size_t)m, false);
        for (int v : p.order) {
            if ((unsigned)v >= (unsigned)m) return fail("order_rid entry out of range");
            if (seen[(size_t)v]) return fail("order_rid is not a permutation: two writers for one y index in natural order");
            seen[(size_t)v] = true;
        }
    }
    if (!p.dst_map.empty()) {
        if (p.dst_map.size() != (size_t)m) return fail("dst_map size");
        for (int v : p.dst_map) if ((unsigned)v >= (unsigned)m) return fail("dst_map entry out of range");
    }
    if (p.stats.rowA != p.m || p.stats.colA != p.n || p.stats.nnzA != p.nnz || p.stats.precision != p.precision) return fail("stats header");
    if (p.stats.row_long < 0 || p.n_mfma_rows < 0 || p.n_mfma_rows > p.stats.row_block || (long long)p.stats.row_long + p.stats.row_block > m) return fail("category counters");
    if (n_panels == 0 && (p.med_slot0 < p.stats.row_long || (long long)p.med_slot0 + p.n_mfma_rows > (long long)p.stats.row_long + p.stats.row_block)) return fail("med_slot0");
    if (p.panel_bounds.size() != 2 * (size_t)n_panels) return fail("panel_bounds size");
    if (p.two_phase) {    // order + stats + the tile streams (twophase.cpp)
        if (n_panels > 0 || is_panel) return fail("two-phase plan with column panels");
        if (p.cnt_long || p.cnt_reg || p.cnt_irr || p.cnt_short || p.cnt_rt || !p.med_ptr.empty() || !p.irr_ptr.empty() || !p.piece_ptr.empty()) return fail("two-phase plan holds packed DASP arrays");
        if (!validate_two_phase(p, why)) return false;
        if (!validate_long_cb(p, 0, why)) return false;          // the hybrid's hub rows (none: empty arrays)
        long long have = 0;                                       // every nonzero once: in a tile stream or in the column-blocked hub rows
        for (uint16_t r : p.tp.lrow) have += r != kTpPadRow;
        for (uint16_t c : p.lcb.lcol) have += c != kLcbPadCol;
        if (have != p.nnz) return fail("the two-phase streams and the column-blocked hub rows do not add up to nnzA");
        return true;
    }
    if (!p.tp.dst.empty() || !p.tp.lcol.empty() || !p.tp.lrow.empty() || !p.tp.val.empty() || !p.tp.unit.empty() || !p.tp.rb_row0.empty() || !p.tp.rb_seg0.empty()) return fail("two-phase streams in a plan that is not two-phase");
    if (n_panels == 0 && (p.lcb.n_rows() > 0 || !p.lcb.ptr.empty() || !p.lcb.lcol.empty() || !p.lcb.val.empty() || !p.lcb.unit.empty() || !p.lcb.row_id.empty())) return fail("column-blocked long rows outside a column-panel plan");
    if (n_panels > 0) {   // a panel parent keeps order + stats only
        if (p.cnt_long || p.cnt_reg || p.cnt_irr || p.cnt_short || !p.med_ptr.empty() || !p.irr_ptr.empty() || !p.piece_ptr.empty()) return fail("panel parent holds packed arrays");
        for (int k = 0; k < n_panels; ++k)
            if (p.panel_bounds[2 * k] < 0 || p.panel_bounds[2 * k + 1] < p.panel_bounds[2 * k] || p.panel_bounds[2 * k + 1] > xlen) return fail("panel_bounds range");
        return validate_long_cb(p, n_panels, why);
    }
    auto cid_ok = [&](const raw_vector<int> &c) { for (int v : c) if (v < -1 || v >= xlen) return false; return true; };
    auto mono = [](const std::vector<int> &a) { if (a.empty() || a[0] != 0) return false; for (size_t i = 1; i < a.size(); ++i) if (a[i] < a[i - 1]) return false; return true; };
    // ---- long rows
    if (p.long_val.size() != p.cnt_long * (size_t)vb || p.long_cid.size() != p.cnt_long) return fail("long arrays");
    if (p.piece_ptr.size() != p.piece_dst.size() + 1 || !mono(p.piece_ptr) || (size_t)p.piece_ptr.back() != p.cnt_long) return fail("piece_ptr");
    if (p.multi_ptr.size() != p.multi_dst.size() + 1 || !mono(p.multi_ptr)) return fail("multi_ptr");
    const long long n_partial = p.multi_ptr.back();
    for (int d : p.piece_dst) if (d >= 0 ? d >= m : (long long)~d >= n_partial) return fail("piece_dst out of range");
    for (int d : p.multi_dst) if ((unsigned)d >= (unsigned)m) return fail("multi_dst out of range");
    for (size_t i = 0; i + 1 < p.piece_ptr.size(); ++i) if (p.piece_ptr[i] % kLongAlign) return fail("piece_ptr alignment");
    if (!cid_ok(p.long_cid)) return fail("long_cid out of range");
    {   // 16-bit ids of the long pieces: the chunk prefix, and -- narrow pieces -- base + offset == the 32-bit id of every element, pads at the pads; wide pieces hold zeros
        const size_t np = p.piece_dst.size();
        if (p.piece_c16.size() != 2 * np || p.long_cid16.size() != p.cnt_long) return fail("long_cid16 / piece_c16 sizes");
        long long chunks = 0;
        for (size_t q = 0; q < np; ++q) {
            if (p.piece_c16[2 * q] != chunks || (unsigned)p.piece_c16[2 * q + 1] > 1u) return fail("piece_c16");
            chunks += (p.piece_ptr[q + 1] - p.piece_ptr[q] + CH - 1) / CH;
        }
        if ((long long)p.long_base.size() != chunks || (long long)p.cnt_long_chunks != chunks) return fail("long_base size");
        for (size_t q = 0; q < np; ++q) {
            const bool narrow = p.piece_c16[2 * q + 1] != 0;
            if (narrow && p.piece_ptr[q + 1] - p.piece_ptr[q] < kLong16MinChunks * CH) return fail("narrow piece shorter than four chunks");
            for (long long e = p.piece_ptr[q], c = p.piece_c16[2 * q]; e < p.piece_ptr[q + 1]; e += CH, ++c) {
                const int b = p.long_base[(size_t)c];
                if (b < 0 || b >= std::max<long long>(xlen, 1)) return fail("long_base out of range");
                for (long long j = e; j < std::min<long long>(e + CH, p.piece_ptr[q + 1]); ++j) {
                    const int col = p.long_cid[(size_t)j]; const unsigned o = p.long_cid16[(size_t)j];
                    if (narrow ? (col < 0 ? o != kLongPad16 : (o == kLongPad16 || b + (int)o != col)) : o != 0) return fail("long_cid16 does not match long_cid");
                }
            }
        }
    }
    // ---- medium rows
    const long long nb = (p.n_mfma_rows + kMedRows - 1) / kMedRows;
    if (p.med_ptr.size() != (size_t)nb + 1 || !mono(p.med_ptr) || p.stats.n_med_blocks != nb) return fail("med_ptr size / n_med_blocks");
    if ((size_t)p.med_ptr.back() * (size_t)CH != p.cnt_reg || p.med_val.size() != p.cnt_reg * (size_t)vb) return fail("regular tiles");
    if (p.cid16 ? (p.med_cid16.size() + p.med_cid8.size() != p.cnt_reg || p.med_cid8.size() != p.cnt_reg8 || !p.med_cid.empty() ||
                   p.med_base.size() != (size_t)p.med_ptr.back() || p.med_korig.size() != p.med_base.size() || p.med_c8ptr.size() != (size_t)nb + 1)
                : (p.med_cid.size() != p.cnt_reg || !p.med_cid16.empty() || !p.med_cid8.empty() || !p.med_base.empty() || !p.med_c8ptr.empty() || !p.med_korig.empty()))
        return fail("medium column ids");
    // layout of block b (plan.hpp): npair chunks stored in pairs; in cid16 mode its first n8 positions carry one-byte ids and position q holds chunk korig[q]
    auto npair_of = [&](long long b) {
        const long long r0 = b * kMedRows, K = CH / kMedRows;
        return med_npair((int)(p.med_ptr[(size_t)b + 1] - p.med_ptr[(size_t)b]), (int)((p.irr_ptr[(size_t)r0 + 1] - p.irr_ptr[(size_t)r0] + K - 1) / K), (int)vb, p.pair_mode);
    };
    auto oneshot_of = [&](long long b) {
        const long long r0 = b * kMedRows, K = CH / kMedRows;
        return vb == 8 && med_oneshot64((int)(p.med_ptr[(size_t)b + 1] - p.med_ptr[(size_t)b]), (int)((p.irr_ptr[(size_t)r0 + 1] - p.irr_ptr[(size_t)r0] + K - 1) / K));
    };
    if (p.irr_ptr.size() != (size_t)p.n_mfma_rows + 1 || !mono(p.irr_ptr) || (size_t)p.irr_ptr.back() != p.cnt_irr) return fail("irr_ptr");
    if (p.irr_val.size() != p.cnt_irr * (size_t)vb || p.irr_cid.size() != p.cnt_irr) return fail("irregular arrays");
    if (!cid_ok(p.irr_cid) || !cid_ok(p.med_cid)) return fail("medium column id out of range");
    if (p.cid16) {
        if (!mono(p.med_c8ptr) || (size_t)p.med_c8ptr.back() * (size_t)CH != p.cnt_reg8) return fail("med_c8ptr");
        for (long long b = 0; b < nb; ++b) {
            const int nc = p.med_ptr[(size_t)b + 1] - p.med_ptr[(size_t)b], n8 = p.med_c8ptr[(size_t)b + 1] - p.med_c8ptr[(size_t)b], npair = npair_of(b);
            const bool one = med_oneshot64(nc, (p.irr_ptr[(size_t)b * kMedRows + 1] - p.irr_ptr[(size_t)b * kMedRows] + 3) / 4);
            if (n8 < 0 || n8 > npair || n8 % (one ? 2 : kMedBatch64) || (n8 && vb != 8))
                return fail("med_c8ptr: one-byte ids come in whole batches (pipelined) / pairs (one-shot) of an f64 block's paired region");
            std::vector<char> seen((size_t)nc, 0);
            for (int q = 0; q < nc; ++q) {
                const unsigned k = (unsigned)p.med_korig[(size_t)p.med_ptr[(size_t)b] + (size_t)q];
                if (k >= (unsigned)nc || seen[k] || (q >= npair && (int)k != q)) return fail("med_korig is not a permutation of its block's paired region");
                seen[k] = 1;
            }
        }
    }
    // column id of element e (= lane * vpl + j) of the chunk at position c (absolute) of block b, -1 = pad
    const int vpl = (int)(CH / 64);
    auto cid_at = [&](long long b, long long c, long long e) -> long long {
        const long long c0 = p.med_ptr[(size_t)b];
        const int npair = npair_of(b), q = (int)(c - c0), lane = (int)(e / vpl), j = (int)(e % vpl);
        if (!p.cid16) return p.med_cid[(size_t)c0 * (size_t)CH + med_elem_index(npair, q, lane, j, vpl, (int)CH)];
        const long long c8 = p.med_c8ptr[(size_t)b];
        const int n8 = (int)(p.med_c8ptr[(size_t)b + 1] - c8);
        if (q < n8) { const unsigned o = p.med_cid8[(size_t)c8 * (size_t)CH + med_cid8_index(q, lane, (int)CH, oneshot_of(b))]; return o == 0xFFu ? -1 : (long long)p.med_base[(size_t)c] + o; }
        const unsigned o = p.med_cid16[(size_t)(c0 - c8) * (size_t)CH + med_elem_index(npair - n8, q - n8, lane, j, vpl, (int)CH)];
        return o == 0xFFFFu ? -1 : (long long)p.med_base[(size_t)c] + o;
    };
    if (p.cid16)
        for (long long b = 0; b < nb; ++b)
            for (long long c = p.med_ptr[(size_t)b]; c < p.med_ptr[(size_t)b + 1]; ++c) {
                const long long base = p.med_base[(size_t)c];
                if (base < 0 || base >= std::max<long long>(xlen, 1)) return fail("med_base out of range");
                for (long long e = 0; e < CH; ++e) if (cid_at(b, c, e) >= xlen) return fail("16-bit column id out of range");
            }
    // ---- windows
    if (p.lds_bytes < 0 || p.lds_bytes > kWinLdsMax) return fail("lds_bytes");
    if (p.windowed) {
        if (p.row_window < 64 || p.row_window > 1024 || p.row_window % kMedRows) return fail("row_window");
        const long long nW = (p.n_mfma_rows + p.row_window - 1) / p.row_window;
        if (p.win_len.size() != (size_t)nW || p.win_cmin.size() != (size_t)nW || p.med_dst.size() != (size_t)p.n_mfma_rows) return fail("window tables");
        for (int d : p.med_dst) if ((unsigned)d >= (unsigned)m) return fail("med_dst out of range");
        const long long A = 16 / vb, bpw = p.row_window / kMedRows;
        for (long long w = 0; w < nW; ++w) {
            const long long len = p.win_len[(size_t)w], c0 = p.win_cmin[(size_t)w];
            if (len < 0 || c0 < 0 || len * vb > p.lds_bytes || c0 % A || (len > 0 && c0 + len > xlen)) return fail("window span");
            if (len == 0 || p.win_hybrid) continue;     // hybrid: gathers outside the span read global memory (ids already range-checked)
            // every gather of an LDS-staged window must fall inside the staged span
            for (long long b = w * bpw; b < std::min(nb, (w + 1) * bpw); ++b) {
                for (long long c = p.med_ptr[(size_t)b]; c < p.med_ptr[(size_t)b + 1]; ++c)
                    for (long long e = 0; e < CH; ++e) {
                        const long long col = cid_at(b, c, e);
                        if (col < 0) continue;
                        if (col < c0 || col >= c0 + len) return fail("windowed column id outside its LDS span");
                    }
                for (long long r = b * kMedRows; r < std::min<long long>(p.n_mfma_rows, (b + 1) * kMedRows); ++r)
                    for (int t = p.irr_ptr[(size_t)r]; t < p.irr_ptr[(size_t)r + 1]; ++t) { const int col = p.irr_cid[(size_t)t]; if (col >= 0 && (col < c0 || col >= c0 + len)) return fail("windowed tail id outside its LDS span"); }
            }
        }
    } else if (!p.med_dst.empty() || !p.win_len.empty() || !p.win_cmin.empty()) return fail("window tables without windows");
    // ---- short rows / slabs
    if (p.short_val.size() != p.cnt_short * (size_t)vb || p.short_cid.size() != p.cnt_short || !cid_ok(p.short_cid)) return fail("short arrays");
    long long off = 0, tile0 = 0;
    static const int kLen[5] = {1, 2, 3, 4, 0};
    for (int g = 0; g < kNumShortGroups; ++g) {
        const ShortGroup &G = p.grp[g];
        if (G.len != (g < 5 ? kLen[g] : g) || G.count < 0 || G.count > m) return fail("short group length / count");
        const bool seg_ok = G.seg == 0 || (G.seg == 1 && G.len >= 1 && G.len <= 4 && !p.windowed);
        if (!seg_ok || G.rpt != (G.seg ? short_seg_rows(G.len) : (int)SR)) return fail("short group layout");
        if (G.tiles != (G.count + G.rpt - 1) / G.rpt || G.tile0 != tile0 || G.elem_off != off) return fail("short group tiles / offsets");
        tile0 += G.tiles; off += (long long)G.tiles * short_tile_elems(G.seg != 0, G.len, (int)SR);
        const SlotMap &M = G.map;
        if (M.split < 0 || M.grp[0] < 0 || M.grp[1] < 0) return fail("slot map");
        for (int t : {0, M.split - 1, M.split, G.count - 1})
            if (t >= 0 && t < G.count) { const long long sl = M.slot(t); if (sl < 0 || sl >= m) return fail("slot map leaves the permutation"); }
    }
    if ((size_t)off != p.cnt_short || p.stats.n_short_tiles != tile0) return fail("short segment size");
    // ---- row tiles (a column panel only): what row_tile() indexes with
    if (p.rt_max == 0) { if (!p.rt_ptr.empty() || !p.rt_start.empty() || !p.rt_mask.empty() || p.cnt_rt) return fail("row tiles without rt_max"); }
    else {
        const size_t tiles = (size_t)((m + kRowTile - 1) / kRowTile);
        if (p.rt_max < 0 || p.rt_max > kRowTileMax || !is_panel || p.windowed || p.cnt_reg8) return fail("row tiles: bound / not a plain column panel");
        if (p.rt_mask.size() != tiles || p.rt_ptr.size() != tiles + 1 || p.rt_start.size() != tiles * kRowTile || !mono(p.rt_ptr) || (size_t)p.rt_ptr.back() != p.cnt_rt)
            return fail("row tile tables");
        if (p.rt_val.size() != p.cnt_rt * (size_t)vb || p.rt_cid.size() != p.cnt_rt) return fail("row tile arrays");
        for (int v : p.rt_cid) if (v < 0 || v >= xlen) return fail("row tile column id out of range");
        for (size_t t = 0; t < tiles; ++t) {
            const int n = p.rt_ptr[t + 1] - p.rt_ptr[t];
            for (int i = 0; i < kRowTile; ++i) {
                const int s = p.rt_start[t * kRowTile + (size_t)i], e = i + 1 < kRowTile ? p.rt_start[t * kRowTile + (size_t)i + 1] : n;
                const bool on = (p.rt_mask[t] >> i) & 1;
                if ((i == 0 && s != 0) || e < s || e > n || e - s > p.rt_max || (!on && e != s) || (on && (long long)t * kRowTile + i >= m)) return fail("row tile row table");
            }
        }
    }
    if (p.win_hybrid && !p.windowed) return fail("win_hybrid without windows");
    if (p.win_rel16) {       // the kernel then takes win_cmin as EVERY chunk's base in an LDS-staged window, without reading med_base
        if (!p.windowed || !p.cid16 || p.win_hybrid) return fail("win_rel16 needs LDS windows with 16-bit ids");
        const long long bpw = p.row_window / kMedRows;
        for (long long b = 0; b < nb; ++b) {
            const size_t w = (size_t)(b / bpw);
            if (p.win_len[w] <= 0) continue;
            if (p.win_len[w] > 65534) return fail("win_rel16: staged span beyond 16-bit offsets");
            for (long long c = p.med_ptr[(size_t)b]; c < p.med_ptr[(size_t)b + 1]; ++c)
                if (p.med_base[(size_t)c] != p.win_cmin[w]) return fail("win_rel16: a chunk base differs from its window's first staged column");
        }
    }
    // ---- one writer per y index: put_y is a plain store (or, in accumulate mode, a plain read-modify-write), so two units with the
    // same destination would race / add twice.  Destinations exactly as the kernels form them (upload_plan: order / dst_map).
    {
        const bool natural = p.opt.y_order == DASP_Y_NATURAL;
        std::vector<bool> hit((size_t)m, false);
        auto claim = [&](long long yi) { if (yi < 0 || yi >= m || hit[(size_t)yi]) return false; hit[(size_t)yi] = true; return true; };
        auto ydst = [&](long long slot) -> long long {
            if (slot < 0 || slot >= m) return -1;
            if (!natural) return slot;
            const int r = p.order[(size_t)slot];
            return p.dst_map.empty() ? r : p.dst_map[(size_t)r];
        };
        for (size_t t = 0; t < p.rt_mask.size(); ++t)      // a row tile stores its positions of the output order directly
            for (int i = 0; i < kRowTile; ++i) if (((p.rt_mask[t] >> i) & 1) && !claim((long long)t * kRowTile + i)) return fail("two writers for one y index (row tiles)");
        for (int d : p.piece_dst) if (d >= 0 && !claim(d)) return fail("two writers for one y index (piece_dst)");
        for (int d : p.multi_dst) if (!claim(d)) return fail("two writers for one y index (multi_dst)");
        for (long long r = 0; r < p.n_mfma_rows; ++r)
            if (!claim(p.windowed ? (long long)p.med_dst[(size_t)r] : ydst((long long)p.med_slot0 + r))) return fail("two writers for one y index (medium rows)");
        for (int g = 0; g < kNumShortGroups; ++g) {
            if (is_panel && p.grp[g].len == 0) continue;       // a panel never stores its empty rows (DevArgs::skip0; the tiled rows are among them)
            for (int t = 0; t < p.grp[g].count; ++t)
                if (!claim(ydst(g < 5 ? (long long)p.grp[g].map.slot(t) : (long long)p.grp[g].map.base[0] + t))) return fail("two writers for one y index (short rows / slabs)");
        }
    }
    return true;
}

static void write_plan(Writer &w, Plan &p)
{
    const int hdr[25] = {p.precision, p.m, p.n, p.nnz, p.opt.y_order, p.windowed ? 1 : 0, p.row_window, p.lds_bytes, p.cid16 ? 1 : 0,
                         p.opt.n_parts, p.opt.part_stride, p.opt.stream_policy, (int)p.panels.size(), p.n_mfma_rows, p.win_hybrid ? 1 : 0, p.med_slot0, p.pair_mode, p.win_rel16 ? 1 : 0, p.rt_max,
                         p.two_phase ? 1 : 0, p.tp.cb, p.tp.rb_max, p.lcb.cb, p.lcb.n_cb, p.lcb.h};
    w.raw(hdr, sizeof hdr); w.raw(&p.stats, sizeof p.stats); w.raw(p.grp, sizeof p.grp);
    arrays(w, p);
    for (auto &h : p.panels) write_plan(w, h->impl);
}

static bool read_plan(Reader &r, Plan &p, int depth, std::string &r_why)
{
    int hdr[25];
    r.raw(hdr, sizeof hdr);
    if (!r.ok || (hdr[0] != 64 && hdr[0] != 16) || hdr[12] < 0 || hdr[12] > 64 || (depth > 0 && hdr[12] != 0)) return false;
    p.precision = hdr[0]; p.geo = geometry_for(p.precision);
    p.m = hdr[1]; p.n = hdr[2]; p.nnz = hdr[3];
    dasp_options_default(&p.opt);
    p.opt.y_order = hdr[4]; p.windowed = hdr[5] != 0; p.row_window = hdr[6]; p.lds_bytes = hdr[7]; p.cid16 = hdr[8] != 0;
    p.opt.n_parts = hdr[9]; p.opt.part_stride = hdr[10]; p.opt.stream_policy = hdr[11]; p.n_mfma_rows = hdr[13]; p.win_hybrid = hdr[14] != 0; p.med_slot0 = hdr[15]; p.pair_mode = hdr[16]; p.win_rel16 = hdr[17] != 0; p.rt_max = hdr[18];
    if (p.pair_mode < 0 || p.pair_mode > 2 || (p.windowed && p.pair_mode)) return false;
    p.two_phase = hdr[19] != 0; p.tp.cb = hdr[20]; p.tp.rb_max = hdr[21];
    p.lcb.cb = hdr[22]; p.lcb.n_cb = hdr[23]; p.lcb.h = hdr[24];
    if (p.two_phase && (depth > 0 || hdr[12] != 0)) return false;
    r.raw(&p.stats, sizeof p.stats); r.raw(p.grp, sizeof p.grp);
    arrays(r, p);
    if (!r.ok) return false;
    p.tp.segments = p.tp.dst.size();
    p.lcb.elems = p.lcb.lcol.size();
    p.opt.two_phase = p.two_phase ? 1 : -1; p.opt.tp_col_block = p.tp.cb; p.opt.tp_row_block = p.tp.rb_max;
    p.opt.part_bounds = p.part_bounds.empty() ? nullptr : p.part_bounds.data();
    const size_t vb = (size_t)p.geo.vbytes;
    const int np = hdr[12];
    bool sane = p.m >= 0 && p.order.size() == (size_t)p.m && p.stats.rowA == p.m && p.stats.precision == p.precision &&
                (p.dst_map.empty() || p.dst_map.size() == (size_t)p.m) && p.panel_bounds.size() == 2 * (size_t)np &&
                p.long_val.size() == p.long_cid.size() * vb && p.irr_val.size() == p.irr_cid.size() * vb &&
                p.short_val.size() == p.short_cid.size() * vb &&
                (p.cid16 ? p.med_val.size() == (p.med_cid16.size() + p.med_cid8.size()) * vb : p.med_val.size() == p.med_cid.size() * vb);
    if (np == 0 && !p.two_phase)   // a packed plan (a panel parent and a two-phase plan keep none of the row-structure arrays)
        sane = sane && p.piece_ptr.size() == p.piece_dst.size() + 1 && p.irr_ptr.size() == (size_t)p.n_mfma_rows + 1 &&
               p.n_mfma_rows >= 0 && p.n_mfma_rows <= p.stats.row_block && (!p.windowed || p.med_dst.size() == (size_t)p.n_mfma_rows);
    if (!sane) return false;
    p.cnt_long = p.long_cid.size(); p.cnt_irr = p.irr_cid.size(); p.cnt_short = p.short_cid.size();
    p.cnt_long_chunks = p.long_base.size();
    p.cnt_reg = p.cid16 ? p.med_cid16.size() + p.med_cid8.size() : p.med_cid.size();
    p.cnt_reg8 = p.med_cid8.size();
    p.cnt_rt = p.rt_cid.size();
    p.host_dropped = false;
    p.panel = depth > 0;
    if (!validate_plan(p, np, depth > 0, r_why)) return false;
    p.opt.col_panels = np > 0 ? np : 1;
    for (int k = 0; k < np; ++k) {
        std::unique_ptr<dasp_plan> h(new dasp_plan());
        if (!read_plan(r, h->impl, depth + 1, r_why) || h->impl.m != p.m || h->impl.precision != p.precision) return false;
        p.panels.push_back(std::move(h));
    }
    if (np > 0) {      // every nonzero once: in a panel, or in the column-blocked long rows
        long long have = 0;
        for (const auto &h : p.panels) have += (long long)h->impl.nnz + (long long)h->impl.cnt_rt;      // (a panel's row tiles are not in its nnz)
        for (uint16_t c : p.lcb.lcol) have += c != kLcbPadCol;
        if (have != p.nnz) { r_why = "the panels and the column-blocked long rows do not add up to nnzA"; return false; }
    }
    return true;
}

int save_plan(Plan &p, const char *path)
{
    if (!path) return DASP_ERR_ARG;
    if (p.host_dropped) { set_error("host arrays were dropped: nothing to save"); return DASP_ERR_STATE; }
    FILE *f = std::fopen(path, "wb");
    if (!f) { set_error(std::string("cannot create ") + path); return DASP_ERR_OPEN; }
    Writer w{f};
    w.raw(kPlanMagic, 8);
    const int abi[3] = {(int)sizeof(dasp_stats_t), kNumShortGroups, (int)sizeof(ShortGroup)};
    w.raw(abi, sizeof abi);
    write_plan(w, p);
    const bool ok = (std::fclose(f) == 0) && w.ok;
    if (!ok) { set_error(std::string("short write to ") + path); return DASP_ERR_OPEN; }
    return DASP_OK;
}

int load_plan(Plan &p, const char *path)
{
    if (!path) return DASP_ERR_ARG;
    FILE *f = std::fopen(path, "rb");
    if (!f) { set_error(std::string("cannot open ") + path); return DASP_ERR_OPEN; }
    std::fseek(f, 0, SEEK_END);
    const long long size = std::ftell(f);
    std::fseek(f, 0, SEEK_SET);
    Reader r{f, size < 0 ? 0 : size};
    char magic[8];
    int abi[3] = {0, 0, 0};
    r.raw(magic, 8);
    r.raw(abi, sizeof abi);
    if (!r.ok || std::memcmp(magic, kPlanMagic, 8) != 0 || abi[0] != (int)sizeof(dasp_stats_t) || abi[1] != kNumShortGroups || abi[2] != (int)sizeof(ShortGroup)) {
        std::fclose(f); set_error("not a DASPPLN7 plan file of this build (magic / struct sizes differ)"); return DASP_ERR_BANNER;
    }
    std::string why;
    bool ok = false;
    try { ok = read_plan(r, p, 0, why); }
    catch (...) { std::fclose(f); throw; }
    std::fclose(f);
    if (!ok) { set_error("truncated or inconsistent plan file" + (why.empty() ? std::string() : ": " + why)); return DASP_ERR_ENTRY; }
    return DASP_OK;
}

}  // namespace dasp

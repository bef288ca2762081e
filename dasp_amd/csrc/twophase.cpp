// twophase.cpp -- the two-phase (gather-free) form of a plan: the automatic rule, the host packer and the checks of a loaded plan file.
// Layout and kernels: plan.hpp (struct TwoPhase), kernels.hip (dasp_tp_expand_kernel / dasp_tp_reduce_kernel), DESIGN.md section 4.7.
// No reference counterpart: the reference's kernels gather x per nonzero (src/dasp_f16.h:133-590); this is the path for matrices on which that
// gather, not HBM, is the bound.
#include <algorithm>
#include <atomic>
#include <chrono>
#include <cstdint>
#include <cstring>
#include <string>
#include <thread>
#include <vector>

#include "plan.hpp"

namespace dasp {

namespace {
template <class F>
void for_each_block(int n, int threads, F f)
{
    threads = std::max(1, std::min(threads, n));
    if (threads == 1) { for (int i = 0; i < n; ++i) f(i); return; }
    std::atomic<int> next{0};
    std::vector<std::thread> th;
    auto work = [&] { for (int i = next++; i < n; i = next++) f(i); };
    for (int t = 1; t < threads; ++t) th.emplace_back(work);
    work();
    for (auto &t : th) t.join();
}
}  // namespace

// the automatic rule (opt.two_phase == 0): f16, identity column map, no explicit choice of column panels, >= 10 M nonzeros in rows whose gathers scatter over x
// (`scattered`: decide_panels' samples -- > 50 % of a row's nonzeros on distinct 128-byte lines, a third of the entries in rows spanning > x / 4).  Neither the size of x nor hot lines
// matter to this form: powerlaw_1M f16 (x = 2 MB) 0.324 -> 0.209 ms, rmat_2M f16 (92 % of the gathers on 3 MiB of hot lines) 0.135 -> 0.094.  Hub rows are fine since
// phase 2 combines a lane's consecutive same-row elements before the atomic (before that: rmat_2M 3.13 ms); small matrices lose (webbase-1M, 3.6 M nonzeros:
// 0.0149 -> 0.0246 ms: two launches and a slice of x per workgroup), hence the size bound (decide_panels: 10 M nonzeros since the r5 size sweep; 16 M before it).
int decide_two_phase(const Plan &p, const int *rp, int scattered)
{
    (void)rp;
    if (p.panel || p.opt.two_phase < 0) return 0;
    if (p.opt.two_phase > 0) return 1;                              // forced: build_two_phase refuses what it cannot do
    if (p.precision != 16 || p.opt.n_parts > 0 || !p.dst_map.empty()) return 0;
    if (p.opt.col_panels >= 2 || p.opt.col_panels == 1 || p.opt.col_panels < 0) return 0;      // (an explicit choice about column panels is a choice of form)
    return scattered ? 1 : 0;
}

int build_two_phase(Plan &p, const int *rp, const int *ci, const void *val, const unsigned char *skip)
{
    // skip[row] != 0 (r6, the f16 hybrid): the row's nonzeros are not this form's -- the hub rows of a plan whose Plan::lcb holds them column-blocked; the row keeps its
    // output position and gets y = 0 from phase 2 (dasp_lcb_reduce_kernel stores / adds its sum behind it)
    auto len_of = [&](int r) { return skip && skip[r] ? 0 : rp[r + 1] - rp[r]; };
    using clk = std::chrono::steady_clock;
    const auto t_begin = clk::now();
    if (p.precision != 16) { set_error("two_phase: f16 plans only"); return DASP_ERR_ARG; }
    if (p.opt.n_parts > 0) { set_error("two_phase: not with a column remap (n_parts)"); return DASP_ERR_ARG; }
    const int m = p.m, n = p.n;
    const long long nnz = p.nnz;
    const int vb = p.geo.vbytes;
    int cb = p.opt.tp_col_block > 0 ? p.opt.tp_col_block : kTpColBlock;
    int rbm = p.opt.tp_row_block > 0 ? p.opt.tp_row_block : kTpRowBlock;
    if (cb % 8 || cb < 8 || cb > 65536) { set_error("tp_col_block must be a multiple of 8 in 8 .. 65536"); return DASP_ERR_ARG; }
    if (rbm < 1 || rbm > 8192) { set_error("tp_row_block must be in 1 .. 8192"); return DASP_ERR_ARG; }
    const int threads = resolve_threads(p.opt.host_threads);
    if (p.opt.tp_col_block <= 0 && nnz > 0) {
        // auto: half-size column blocks when the columns are skewed -- one block of 32768 holding several times its share of the nonzeros (R-MAT: the heaviest holds 19 %, 12 x
        // the mean; the other stand-ins 1.0-1.3 x).  rmat_2M f16 0.0893 -> 0.0826 ms; every even family prefers 32768 (fewer tiles, less padding: ljournal-2008 0.1745 / 0.1891,
        // webbase-1M x4 0.0529 / 0.0690: tools/scratch/tp_blocks_sweep2.sh)
        const int nb = std::max(1, (n + kTpColBlock - 1) / kTpColBlock);
        std::vector<long long> hist((size_t)nb, 0);
        const long long stride = std::max<long long>(1, nnz >> 22);              // <= ~4 M samples
        long long taken = 0;
        for (long long j = 0; j < nnz; j += stride) { hist[(size_t)(ci[j] / kTpColBlock)]++; ++taken; }
        const long long heaviest = *std::max_element(hist.begin(), hist.end());
        if (nb >= 8 && heaviest * nb > 4 * taken) cb = kTpColBlock / 2;
    }
    TwoPhase &t = p.tp;
    t = TwoPhase{};
    t.cb = cb; t.rb_max = rbm;
    const bool natural = p.opt.y_order == DASP_Y_NATURAL;
    const int *order = p.order.data();                 // output position -> row (permuted order)
    auto row_at = [&](int pos) { return natural ? pos : order[pos]; };
    const int n_cb = std::max(1, (n + cb - 1) / cb);

    // ---- row blocks: consecutive output positions, <= rbm of them, about `target` nonzeros each (the permuted order is sorted by row
    // length: equal counts of rows would give the first blocks a hundred times the work of the last)
    const long long target = std::max<long long>(16384, std::min<long long>(1 << 17, nnz / 1024 + 1));
    t.rb_row0.push_back(0);
    {
        long long acc = 0; int rows = 0;
        for (int pos = 0; pos < m; ++pos) {
            const int r = row_at(pos);
            const int len = len_of(r);
            if (rows > 0 && (rows >= rbm || acc + len > target)) { t.rb_row0.push_back(pos); acc = 0; rows = 0; }
            acc += len; ++rows;
        }
        if (m > 0) t.rb_row0.push_back(m);
    }
    const int n_rb = t.n_rb();
    // the automatic rule's last word: every non-empty (row block, column block) tile is padded to whole 64-element segments, and a row block holds at most
    // rbm rows -- a large, very sparse matrix (50 M rows of 3 nonzeros: 1526 column blocks, ~8 nonzeros per tile) would store several times its nonzeros and keep
    // n_rb * n_cb * 20 bytes of tables on the host.  Declined (kTpDeclined: build_impl goes on to column panels / the plain plan) when the padded streams pass 3 x the
    // nonzeros (webbase-1M x4, 1.59 x padding, wins 53.0 against 61.5 us; x16, 2.69 x, 291.9 against 328.2 us; the form's time follows its padded size, so ~3 x is the break-even there) or the tile table passes 64 M entries; a forced two_phase = 1 is built as asked.
    const bool may_decline = p.opt.two_phase == 0;
    long long nnz_here = 0;
    for (int r = 0; r < m; ++r) nnz_here += len_of(r);
    if (may_decline && (long long)n_rb * (long long)n_cb > (1ll << 26)) { t = TwoPhase{}; return kTpDeclined; }
    // ---- nonzeros per tile
    std::vector<int> cnt((size_t)n_rb * (size_t)n_cb, 0);
    for_each_block(n_rb, threads, [&](int b) {
        int *c = cnt.data() + (size_t)b * (size_t)n_cb;
        for (int pos = t.rb_row0[(size_t)b]; pos < t.rb_row0[(size_t)b + 1]; ++pos) {
            const int r = row_at(pos);
            for (int j = rp[r], je = rp[r] + len_of(r); j < je; ++j) c[ci[j] / cb]++;
        }
    });
    // ---- segment offsets: off2 = RB-major (row block, then column block), off1 = CB-major (column block, then row block)
    std::vector<long long> off2((size_t)n_rb * (size_t)n_cb + 1), off1((size_t)n_rb * (size_t)n_cb);
    auto segs_of = [&](size_t i) { return (long long)((cnt[i] + kTpSeg - 1) / kTpSeg); };
    {
        long long run = 0;
        for (size_t i = 0; i < cnt.size(); ++i) { off2[i] = run; run += segs_of(i); }
        off2[cnt.size()] = run;
        long long run1 = 0;
        for (int c = 0; c < n_cb; ++c)
            for (int b = 0; b < n_rb; ++b) { const size_t i = (size_t)b * (size_t)n_cb + (size_t)c; off1[i] = run1; run1 += segs_of(i); }
        if (run != run1) { set_error("two_phase: internal offset mismatch"); return DASP_ERR_STATE; }
        if (may_decline && run * kTpSeg > 3 * nnz_here) { t = TwoPhase{}; return kTpDeclined; }
        if (run >= (1ll << 31) / 2) { set_error("two_phase: too many segments for 32-bit segment indices"); return DASP_ERR_ARG; }
        t.segments = (size_t)run;
    }
    const size_t S = t.segments;
    t.rb_seg0.resize((size_t)n_rb + 1);
    for (int b = 0; b <= n_rb; ++b) t.rb_seg0[(size_t)b] = (int)off2[std::min((size_t)b * (size_t)n_cb, cnt.size())];
    try {
        t.lcol.resize(S * kTpSeg); t.lrow.resize(S * kTpSeg); t.val.resize(S * kTpSeg * (size_t)vb); t.dst.resize(S);
    } catch (const std::bad_alloc &) { set_error("out of host memory"); return DASP_ERR_NOMEM; }
    const char *vsrc = static_cast<const char *>(val);
    // ---- fill, one row block at a time: a tile's nonzeros in CSR order (rows in output order), then its pads
    for_each_block(n_rb, threads, [&](int b) {
        std::vector<int> cur((size_t)n_cb, 0);
        const size_t base = (size_t)b * (size_t)n_cb;
        const int pos0 = t.rb_row0[(size_t)b];
        for (int pos = pos0; pos < t.rb_row0[(size_t)b + 1]; ++pos) {
            const int r = row_at(pos);
            for (int j = rp[r], je = rp[r] + len_of(r); j < je; ++j) {
                const int c = ci[j] / cb, k = cur[(size_t)c]++;
                const size_t e2 = (size_t)off2[base + (size_t)c] * kTpSeg + (size_t)k, e1 = (size_t)off1[base + (size_t)c] * kTpSeg + (size_t)k;
                t.lrow[e2] = (uint16_t)(pos - pos0);
                std::memcpy(t.val.data() + e2 * (size_t)vb, vsrc + (size_t)j * (size_t)vb, (size_t)vb);
                t.lcol[e1] = (uint16_t)(ci[j] - c * cb);
            }
        }
        for (int c = 0; c < n_cb; ++c) {
            const size_t i = base + (size_t)c;
            const long long sg = segs_of(i);
            for (long long k = cnt[i]; k < sg * kTpSeg; ++k) {
                const size_t e2 = (size_t)off2[i] * kTpSeg + (size_t)k, e1 = (size_t)off1[i] * kTpSeg + (size_t)k;
                t.lrow[e2] = kTpPadRow; std::memset(t.val.data() + e2 * (size_t)vb, 0, (size_t)vb); t.lcol[e1] = 0;
            }
            for (long long s = 0; s < sg; ++s) t.dst[(size_t)(off1[i] + s)] = (int)(off2[i] + s);
        }
    });
    // ---- phase-1 units: runs of <= kTpUnitSegs segments inside one column block
    for (int c = 0; c < n_cb; ++c) {
        const long long s0 = n_rb > 0 ? off1[(size_t)c] : 0;
        const long long s1 = c + 1 < n_cb ? (n_rb > 0 ? off1[(size_t)c + 1] : 0) : (long long)S;
        for (long long s = s0; s < s1; s += kTpUnitSegs) { t.unit.push_back(c); t.unit.push_back((int)s); t.unit.push_back((int)std::min<long long>(s1, s + kTpUnitSegs)); }
    }
    p.two_phase = true;
    p.cnt_long = p.cnt_reg = p.cnt_irr = p.cnt_short = p.cnt_rt = 0;
    // ---- counters: the classifier's stay (whole matrix); the native sizes describe this form
    dasp_stats_t &s = p.stats;
    s.fill0_nnz_short = s.fill0_nnz_long = 0; s.fill0_nnz_reg = (long long)S * kTpSeg;
    s.n_med_blocks = s.n_long_pieces = s.n_long_multi = s.n_short_tiles = 0;
    s.n_workgroups = t.n_units() + n_rb;
    s.x_window_on = s.n_windows = s.n_windows_lds = s.lds_bytes = s.row_window = s.cid16_on = s.x_window_hybrid = s.chunk_pairs = s.cid8_chunks = 0;
    s.med_rows_as_pieces = s.short_seg = s.row_tile_max = s.n_row_tiles = 0; s.row_tile_nnz = 0; s.n_col_panels = 0; s.window_nnz_frac = 0.0;
    s.rate_fill0 = nnz > 0 ? (double)((long long)S * kTpSeg + (long long)p.lcb.elems - nnz) / (double)nnz : 0.0;
    // streamed per SpMV: local columns + xs written (phase 1), values + local rows + xs read (phase 2), the segment map, every unit's slice of x, y
    s.data_X = (long long)S * kTpSeg * (2 + vb + vb + 2 + vb) + (long long)S * 4 + (long long)t.n_units() * (long long)std::min(cb, n) * vb + (long long)m * vb;
    s.two_phase = 1; s.tp_col_block = cb; s.tp_row_blocks = n_rb; s.tp_units = t.n_units(); s.tp_segments = (long long)S; s.tp_seg_elems = kTpSeg;
    s.pre_ms = std::chrono::duration<double, std::milli>(clk::now() - t_begin).count() + s.pre_ms;
    return DASP_OK;
}

// what upload and the kernels rely on, re-derived from the arrays (a plan file is not trusted more than a caller's CSR)
bool validate_two_phase(const Plan &p, std::string &why)
{
    auto fail = [&](const char *w) { why = w; return false; };
    const TwoPhase &t = p.tp;
    if (p.precision != 16) return fail("two-phase plan that is not f16");
    if (p.opt.n_parts > 0 || !p.dst_map.empty()) return fail("two-phase plan with a column remap / destination map");
    if (t.cb % 8 || t.cb < 8 || t.cb > 65536 || t.rb_max < 1 || t.rb_max > 8192) return fail("two-phase block sizes");
    const long long S = (long long)t.segments;
    if (S < 0 || S >= (1ll << 30)) return fail("two-phase segment count");
    if (t.lcol.size() != (size_t)S * kTpSeg || t.lrow.size() != (size_t)S * kTpSeg || t.val.size() != (size_t)S * kTpSeg * (size_t)p.geo.vbytes || t.dst.size() != (size_t)S)
        return fail("two-phase stream sizes");
    const int n_rb = t.n_rb();
    if (p.m == 0 ? !(t.rb_row0.size() <= 1) : (t.rb_row0.size() < 2 || t.rb_row0.front() != 0 || t.rb_row0.back() != p.m)) return fail("rb_row0 span");
    if (t.rb_seg0.size() != (size_t)n_rb + 1 || (n_rb >= 0 && !t.rb_seg0.empty() && (t.rb_seg0.front() != 0 || t.rb_seg0.back() != S))) return fail("rb_seg0 span");
    for (int b = 0; b < n_rb; ++b) {
        const int rows = t.rb_row0[(size_t)b + 1] - t.rb_row0[(size_t)b];
        if (rows < 1 || rows > t.rb_max || t.rb_seg0[(size_t)b + 1] < t.rb_seg0[(size_t)b]) return fail("row block range");
        for (size_t e = (size_t)t.rb_seg0[(size_t)b] * kTpSeg; e < (size_t)t.rb_seg0[(size_t)b + 1] * kTpSeg; ++e)
            if (t.lrow[e] != kTpPadRow && (int)t.lrow[e] >= rows) return fail("local row beyond its row block");
    }
    // dst: a bijection of the segments
    {
        std::vector<bool> seen((size_t)S, false);
        for (int d : t.dst) { if (d < 0 || d >= S || seen[(size_t)d]) return fail("dst is not a permutation of the segments"); seen[(size_t)d] = true; }
    }
    if (t.unit.size() % 3) return fail("unit table");
    const int n_cb = std::max(1, (p.n + t.cb - 1) / t.cb);
    long long at = 0;
    for (int u = 0; u < t.n_units(); ++u) {
        const int c = t.unit[3 * (size_t)u], s0 = t.unit[3 * (size_t)u + 1], s1 = t.unit[3 * (size_t)u + 2];
        if (c < 0 || c >= n_cb || s0 != at || s1 <= s0 || s1 - s0 > kTpUnitSegs || s1 > S) return fail("unit range");
        if (u > 0 && c < t.unit[3 * (size_t)u - 3]) return fail("units not in column-block order");
        const int width = std::min(t.cb, p.n - c * t.cb);
        for (size_t e = (size_t)s0 * kTpSeg; e < (size_t)s1 * kTpSeg; ++e) if ((int)t.lcol[e] >= width) return fail("local column beyond its column block");
        at = s1;
    }
    if (at != S) return fail("units do not cover the segments");
    if (p.stats.two_phase != 1 || p.stats.tp_segments != S || p.stats.tp_row_blocks != n_rb || p.stats.tp_units != t.n_units() || p.stats.tp_col_block != t.cb || p.stats.tp_seg_elems != kTpSeg) return fail("two-phase counters");
    return true;
}

}  // namespace dasp

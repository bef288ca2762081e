// plan.cpp -- row classifier and the long / medium / short packers (host, multi-threaded).
// Mirrors the host half of the reference's spmv_all (src/dasp_f64.h:499-1157,
// src/dasp_f16.h:1029-1443) in the gfx950 geometry described in plan.hpp / DESIGN.md.
#include "plan.hpp"

#include <algorithm>
#include <array>
#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <condition_variable>
#include <functional>
#include <mutex>
#include <thread>

namespace dasp {

static thread_local std::string g_err;
void set_error(const std::string &s) { g_err = s; }
const char *last_error_cstr() { return g_err.c_str(); }

int resolve_threads(int requested)
{
    if (requested > 0) return requested;
    unsigned hc = std::thread::hardware_concurrency();
    if (hc == 0) hc = 1;
    // DASP_HOST_THREADS_MAX: cap of the automatic choice (packing is a memory-bound copy: beyond the host's memory bandwidth more threads add nothing)
    unsigned cap = 32u;     // measured on a 256-core host: HV15R 318 / 204 / 300 / 432 ms at 16 / 32 / 64 / 128 threads
    if (const char *e = std::getenv("DASP_HOST_THREADS_MAX")) cap = (unsigned)std::max(1, std::atoi(e));
    return (int)std::min(hc, cap);
}

// A/B knobs of the packers (tools/*_ab.py): read ONCE per process, before any worker thread exists, and only ever applied over an option
// left at "auto" -- a plan file or a saved number never depends on the environment behind an explicit option.
//   DASP_ONE_PIECE_MAX (longest row kept as one piece, default 4096), DASP_ROW_TILE_MAX (bound of the column panels' row tiles over
//   row_tile_max = 0), DASP_PANEL_ROW_SCAN (the r3 scan order of the f16 panels)
namespace {
struct AbKnobs { int one_piece_max = 4096, row_tile_max = -1; bool panel_row_scan = false; };
const AbKnobs &ab_knobs()
{
    static const AbKnobs k = [] {
        AbKnobs v;
        if (const char *e = std::getenv("DASP_ONE_PIECE_MAX")) v.one_piece_max = std::atoi(e);
        if (const char *e = std::getenv("DASP_ROW_TILE_MAX")) v.row_tile_max = std::max(0, std::atoi(e));
        v.panel_row_scan = std::getenv("DASP_PANEL_ROW_SCAN") != nullptr;
        return v;
    }();
    return k;
}
}  // namespace

// ---- a small persistent worker pool: spawning 32 threads costs ~1 ms, and building a plan from a device-resident CSR is ~20
// O(rows) loops of a few hundred microseconds each (HV15R: 38 ms of which half was thread creation).  Jobs are ranges handed out
// through one atomic counter; the caller works too.  A call from inside a worker (column panels are built side by side, each
// with loops of its own) or while another job runs falls back to plain threads -- never a nested wait on the pool.
namespace {
class Pool {
public:
    static Pool &get() { static Pool p; return p; }
    // run job(part) for part in [0, parts) on up to `parts` threads; false = pool busy / nested: the caller must do it itself
    bool run(long long parts, const std::function<void(long long)> &job)
    {
        if (in_worker_ || parts <= 1) return false;
        std::unique_lock<std::mutex> own(busy_, std::try_to_lock);
        if (!own.owns_lock()) return false;
        ensure((int)std::min<long long>(parts - 1, 127));
        unsigned long long e;
        {
            std::lock_guard<std::mutex> lk(m_);
            job_ = &job; parts_ = parts; next_ = 0; left_ = parts; e = ++epoch_;
        }
        cv_.notify_all();
        work(e);
        std::unique_lock<std::mutex> lk(m_);
        done_.wait(lk, [&] { return left_ == 0; });
        job_ = nullptr;
        return true;
    }
private:
    Pool() = default;
    ~Pool()
    {
        { std::lock_guard<std::mutex> lk(m_); stop_ = true; }
        cv_.notify_all();
        for (auto &t : th_) t.join();
    }
    void ensure(int n)
    {
        while ((int)th_.size() < n) th_.emplace_back([this] { in_worker_ = true; loop(); });
    }
    // parts of job `e` until none is left.  Every piece of job state is read and advanced under m_, and a part is only handed out while
    // the pool is still on job e: a worker that wakes late (after its job is complete and the next one has been set up) must not take
    // an index from the new job's counter on the strength of the old job's bounds -- with a lock-free counter that ran a part twice,
    // counted the job complete one part early and let run() return under a worker still inside the caller's lambda (a rare hang in
    // plan creation, seen twice in ~100 multi-plan probe runs).  A few dozen lock round-trips per job cost nothing next to the loops.
    void work(unsigned long long e)
    {
        for (;;) {
            const std::function<void(long long)> *job;
            long long i;
            {
                std::lock_guard<std::mutex> lk(m_);
                if (epoch_ != e || job_ == nullptr || next_ >= parts_) return;
                i = next_++; job = job_;
            }
            (*job)(i);
            std::lock_guard<std::mutex> lk(m_);
            if (--left_ == 0) done_.notify_all();
        }
    }
    void loop()
    {
        unsigned long long seen = 0;
        for (;;) {
            {
                std::unique_lock<std::mutex> lk(m_);
                cv_.wait(lk, [&] { return stop_ || (epoch_ != seen && job_ != nullptr); });
                if (stop_) return;
                seen = epoch_;
            }
            work(seen);
        }
    }
    std::mutex m_, busy_;
    std::condition_variable cv_, done_;
    std::vector<std::thread> th_;
    const std::function<void(long long)> *job_ = nullptr;
    long long parts_ = 0, left_ = 0, next_ = 0;
    unsigned long long epoch_ = 0;
    bool stop_ = false;
    static thread_local bool in_worker_;
};
thread_local bool Pool::in_worker_ = false;
}  // namespace

// f(begin, end) over [0, n) in contiguous ranges
template <class F>
static void parallel_for(long long n, int threads, long long grain, F f)
{
    if (n <= 0) return;
    long long parts = std::min<long long>(threads, (n + grain - 1) / grain);
    if (parts <= 1) { f(0LL, n); return; }
    const std::function<void(long long)> job = [&](long long t) { f(n * t / parts, n * (t + 1) / parts); };
    if (Pool::get().run(parts, job)) return;
    std::vector<std::thread> th;
    th.reserve((size_t)parts);
    for (long long t = 0; t < parts; ++t) {
        long long b = n * t / parts, e = n * (t + 1) / parts;
        th.emplace_back([=] { f(b, e); });
    }
    for (auto &x : th) x.join();
}

static inline int ceil_div(int a, int b) { return (a + b - 1) / b; }

namespace {
struct Remap {
    int n_parts = 0, stride = 0;
    const int *b = nullptr;
    inline int operator()(int c) const
    {
        if (n_parts <= 0) return c;
        const int *it = std::upper_bound(b, b + n_parts + 1, c);  // first bound > c
        int g = (int)(it - b) - 1;
        return g * stride + (c - b[g]);
    }
};
}  // namespace

// default longest slab-stored medium row (rowlen_probe: blocks reach the slabs' efficiency only past these lengths)
constexpr int kSlabDefaultF64 = 16, kSlabDefaultF16 = 24;

enum BuildMode { kTop = 0, kMetaOnly = 1, kPanel = 2 };   // whole plan (may choose column panels) / order+stats only / one panel

template <class T> static int build_panels(Plan &p, const int *rp, const int *ci, const T *val, int P, const DevCsr *dev);
static int decide_panels(const Plan &p, const int *rp, const int *ci, const Remap &remap, const DevCsr *dev, int *scattered = nullptr);

// ---- the REFERENCE's geometry on the same input (VERDICT r3 missing #5).  The native tiles are 16 x K on 64-lane waves, so the padded sizes
// this library stores differ from the ones the CUDA reference computes (8-row blocks, 8 x 4 tiles, 32-lane warps) -- and a CSV row written
// here could not be compared with a row the reference writes.  All of the reference's sizes are functions of the row lengths alone, so they
// are computed next to the native ones: short tiles dasp_f64.h:609-629 / dasp_f16.h:1139-1156, long rows :1000-1014 / :1273-1288, the
// regular / irregular split :1044-1091 / :1317-1365 (on the globally length-sorted medium rows), rate_fill0 and data_X :1159-1166 /
// dasp_f16.h:1448-1455.  lenL: lengths of the long rows; lenM: medium lengths, descending.
struct RefGeometry { long long fill0_short = 0, fill0_long = 0, fill0_reg = 0, data_X = 0; int nnz_irreg = 0, blocknum = 0, warp_number = 0; double rate_fill0 = 0; };
static RefGeometry reference_geometry(bool f16, int m, int n, int nnz, int n1, int c13, int n3, int n4, int n2, const std::vector<int> &lenL,
                                      const raw_vector<int> &lenM, double threshold, int threads)
{
    constexpr int BS = 8, MK = 4, TILE = BS * MK;           // BlockSize, MMA_K, MMA_M * MMA_K (common.h:29-33)
    constexpr int warps = 4, loop_short = 4, loop_long = 2;  // warpNum_short = warpNum_long, loopNum_short, loopNum_long (dasp_f64.h:18-22)
    RefGeometry g;
    const int per13 = warps * (f16 ? 4 : 2), per22 = per13, per34 = warps * loop_short;      // blocks per thread block
    const int tb13 = ceil_div(ceil_div(c13, BS), per13), tb22 = ceil_div(ceil_div((n2 + 1) / 2, BS), per22), tb34 = ceil_div(ceil_div(n3 + n4, BS), per34);
    g.fill0_short = (f16 ? (long long)((n1 + 1) / 2) * 2 : (long long)n1) + (long long)(tb13 * per13 + tb34 * per34 + tb22 * per22) * TILE;
    const int G = TILE * loop_long * (f16 ? 4 : 1);          // elements one warp takes of a long row: 64 / 256
    long long w = 0;
    for (int L : lenL) w += ceil_div(L, G);
    const long long blocks_long = (w + warps - 1) / warps;
    g.warp_number = (int)(blocks_long * warps);
    g.fill0_long = blocks_long * warps * G;
    const int row_block = (int)lenM.size();
    const int rowloop = row_block < 59990 ? 1 : (row_block < 400000 ? 2 : 4);
    const int nb_real = ceil_div(row_block, BS);
    g.blocknum = ceil_div(nb_real, rowloop * 4) * rowloop * 4;
    std::atomic<long long> reg{0}, irr{0};
    parallel_for(nb_real, threads, 1 << 12, [&](long long b0, long long b1) {
        long long r = 0, t = 0;
        for (long long b = b0; b < b1; ++b) {
            const int lo = (int)b * BS, hi = std::min(row_block, lo + BS);
            long long kept = 0;      // elements of the block's regular part
            for (int k = 1;; ++k) {
                int fill = 0;
                for (int i = lo; i < hi; ++i) { const int L = lenM[(size_t)i]; fill += L / MK >= k ? MK : (L / MK == k - 1 ? L % MK : 0); }
                if ((double)fill >= threshold * TILE) { kept += TILE; continue; }
                for (int i = lo; i < hi; ++i) t += std::max(lenM[(size_t)i] - (k - 1) * MK, 0);
                break;
            }
            if (f16) kept = (kept + 4 * TILE - 1) / (4 * TILE) * (4 * TILE);      // dasp_f16.h:1356
            r += kept;
        }
        reg += r; irr += t;
    });
    g.fill0_reg = reg; g.nnz_irreg = (int)irr;
    const long long stored = g.fill0_short + g.fill0_long + g.nnz_irreg + g.fill0_reg;
    g.rate_fill0 = nnz > 0 ? (double)(stored - nnz) / nnz : 0.0;
    const long long sv = f16 ? 2 : 8;
    g.data_X = (long long)(m + n) * sv + g.fill0_long * (sv + 4) + (long long)g.warp_number * sv + (long long)(lenL.size() + 1) * 4 + g.fill0_short * (sv + 4) +
               g.fill0_reg * (sv + 4) + (long long)(g.blocknum + 1) * 4 + (long long)(f16 ? (g.nnz_irreg + 1) / 2 * 2 : g.nnz_irreg) * (sv + 4) +      // f16: fill0_nnz_irreg, dasp_f16.h:1368,1455
               (long long)(row_block + 1) * 4;
    return g;
}

template <class T>
static int build_impl(Plan &p, const int *rp, const int *ci, const T *val, const DevCsr *dev, BuildMode mode = kTop)
{
    using clk = std::chrono::steady_clock;
    const auto t_begin = clk::now();
    const bool verbose = std::getenv("DASP_VERBOSE") != nullptr;
    auto tick = t_begin;
    auto lap = [&](const char *what) { if (verbose) { auto now = clk::now(); std::fprintf(stderr, "[dasp plan] %-28s %.3f s\n", what, std::chrono::duration<double>(now - tick).count()); tick = now; } };
    const Geometry geo = p.geo;
    const int m = p.m, nnz = p.nnz;
    const int threads = resolve_threads(p.opt.host_threads);
    const bool f16 = p.precision == 16;
    const int block_longest = p.opt.block_longest;
    const double threshold = p.opt.threshold;
    Remap remap;
    remap.n_parts = p.opt.n_parts; remap.stride = p.opt.part_stride; remap.b = p.part_bounds.data();
    const bool meta_only = mode == kMetaOnly;
    const bool pack = !dev && !meta_only;          // the nnz-sized arrays are written here (else: on the device / by the panels)

    // ---- validate CSR (the reference trusts its input; an out-of-range column here would be a
    // wild device read, so it is an argument error instead)
    if (rp[0] != 0 || rp[m] != nnz) { set_error("csrRowPtr[0] != 0 or csrRowPtr[rowA] != nnzA"); return DASP_ERR_ARG; }
    if (mode == kTop) {
        std::atomic<int> bad{0};
        parallel_for(m, threads, 1 << 16, [&](long long b, long long e) {
            for (long long i = b; i < e; ++i) if (rp[i + 1] < rp[i]) { bad = 1; return; }
        });
        const int ncol = p.n;
        if (dev) { if (!bad) { const int rc = devpack_validate(p, *dev); if (rc == DASP_ERR_ARG) bad = 2; else if (rc != DASP_OK) return rc; } }   // a HIP failure is not "bad column"
        else parallel_for(nnz, threads, 1 << 18, [&](long long b, long long e) {
            for (long long i = b; i < e; ++i) if ((unsigned)ci[i] >= (unsigned)ncol) { bad = 2; return; }
        });
        if (bad) { set_error(bad == 1 ? "csrRowPtr not monotone" : "column index out of range"); return DASP_ERR_ARG; }
    }

    lap("validate");
    // opt.sort_columns: rows whose columns do not ascend get their (column, value) pairs sorted by column (stable) -- copies; the caller's arrays stay as they are
    raw_vector<int> ci_sorted;
    raw_vector<T> val_sorted;
    DevCsr dev_sorted{nullptr, nullptr, nullptr};
    std::vector<std::shared_ptr<void>> sort_keep;
    if (mode == kTop && p.opt.sort_columns > 0 && nnz > 0) {
        if (dev) {
            int did = 0;
            if (int rc = devpack_sort_columns(p, *dev, sort_keep, &dev_sorted, &did)) return rc;
            if (did) dev = &dev_sorted;
        } else {
            std::atomic<int> unsorted{0};
            parallel_for(m, threads, 1 << 14, [&](long long b, long long e) {
                for (long long i = b; i < e && !unsorted; ++i)
                    for (int j = rp[i] + 1; j < rp[i + 1]; ++j) if (ci[j] < ci[j - 1]) { unsorted = 1; break; }
            });
            if (unsorted) {
                ci_sorted.resize((size_t)nnz); val_sorted.resize((size_t)nnz);
                parallel_for(m, threads, 1 << 12, [&](long long b, long long e) {
                    std::vector<std::pair<int, T>> row;
                    for (long long i = b; i < e; ++i) {
                        const int a0 = rp[i], a1 = rp[i + 1];
                        bool ok = true;
                        for (int j = a0 + 1; j < a1; ++j) if (ci[j] < ci[j - 1]) { ok = false; break; }
                        if (ok) { for (int j = a0; j < a1; ++j) { ci_sorted[(size_t)j] = ci[j]; val_sorted[(size_t)j] = val[j]; } continue; }
                        row.resize((size_t)(a1 - a0));
                        for (int j = a0; j < a1; ++j) row[(size_t)(j - a0)] = {ci[j], val[j]};
                        std::stable_sort(row.begin(), row.end(), [](const std::pair<int, T> &x, const std::pair<int, T> &y) { return x.first < y.first; });
                        for (int j = a0; j < a1; ++j) { ci_sorted[(size_t)j] = row[(size_t)(j - a0)].first; val_sorted[(size_t)j] = row[(size_t)(j - a0)].second; }
                    }
                });
                ci = ci_sorted.data(); val = val_sorted.data();
            }
        }
        lap("sort columns");
    }
    if (mode == kTop) {
        int scattered = 0;
        const int P = decide_panels(p, rp, ci, remap, dev, &scattered);
        if (P < 0) return P;
        if (decide_two_phase(p, rp, scattered)) {
            // the two-phase (gather-free) form: order_rid and the classifier counters of the whole matrix (row lengths only), then the tile streams
            if (int rc = build_impl<T>(p, rp, ci, val, nullptr, kMetaOnly)) return rc;
            lap("whole-matrix meta");
            // the hybrid (r6): hub rows -- the rule of the column-blocked long rows, decide_long_cb -- leave the two-phase streams for
            // Plan::lcb (x staged in LDS per column block, no atomics on one LDS word per hub row); everything else stays two-phase
            auto hybrid_then_tp = [&](const int *hci, const T *hval) -> int {
                std::vector<unsigned char> in_lcb;
                p.lcb = LongCB{};
                if (decide_long_cb(p, rp, 2, in_lcb, 4, 128) > 0) {
                    std::vector<int> slot_of_row;
                    const bool natural = p.opt.y_order == DASP_Y_NATURAL;
                    if (!natural) { slot_of_row.resize((size_t)m); for (int i = 0; i < m; ++i) slot_of_row[(size_t)p.order[(size_t)i]] = i; }
                    const int rc_l = build_long_cb(p, rp, hci, hval, in_lcb, natural ? nullptr : slot_of_row.data());
                    if (rc_l == 1) in_lcb.clear();
                    else if (rc_l) return rc_l;
                    lap("hub rows by column block");
                }
                const int rc_t = build_two_phase(p, rp, hci, hval, in_lcb.empty() ? nullptr : in_lcb.data());
                if (rc_t != DASP_OK) { p.lcb = LongCB{}; return rc_t; }
                if (!in_lcb.empty()) {
                    const LongCB &L = p.lcb;
                    dasp_stats_t &s = p.stats;
                    s.lcb_rows = L.n_rows(); s.lcb_elems = (long long)L.elems; s.lcb_col_block = L.cb; s.lcb_units = L.n_units();
                    s.n_workgroups += L.n_units() + ceil_div(L.n_rows(), kWavesPerWG);
                    s.data_X += (long long)L.elems * (geo.vbytes + 2) + (long long)L.n_units() * (long long)std::min(L.cb, p.n) * geo.vbytes;
                }
                return DASP_OK;
            };
            int rc2;
            if (!dev) { rc2 = hybrid_then_tp(ci, val); lap("two-phase streams"); if (rc2 != kTpDeclined) return rc2; }
            else {
                raw_vector<int> hci((size_t)nnz);
                raw_vector<T> hval((size_t)nnz);
                if (int rc = devpack_fetch_csr(p, *dev, hci.data(), hval.data())) return rc;
                rc2 = hybrid_then_tp(hci.data(), hval.data());
                lap("two-phase streams (device CSR fetched)");
                if (rc2 == DASP_OK) return devpack_finish_panels(p);       // a device-built plan comes back uploaded
                if (rc2 != kTpDeclined) return rc2;
            }
            // declined by the padding guard (automatic rule only): the matrix keeps its DASP form -- column panels where the rule asked for them, else the plain plan below
        }
        if (P >= 2) return build_panels<T>(p, rp, ci, val, P, dev);
    }
    // ---- classifier: same tests in the same order as dasp_f64.h:499-531.  Two passes over row ranges (count, then fill from the
    // ranges' prefix sums): every list comes out in row order, exactly what the reference's serial loop produces.
    int n1 = 0, n2 = 0, n3 = 0, n4 = 0, nz0 = 0, nlong = 0, nmed = 0;
    const int *scan = p.scan_order;      // a column panel: the parent's slot order (Plan::scan_order); everybody else: row order, as the reference's serial loop
    // category of a row: 0 = len 1, 1 = len 3, 2 = len 2, 3 = empty, 4 = len 4, 5 = long, 6 = medium
    auto cat_of = [block_longest](int len) { return len == 1 ? 0 : len == 3 ? 1 : len == 2 ? 2 : len == 0 ? 3 : len == 4 ? 4 : len >= block_longest ? 5 : 6; };
    const int cparts = (int)std::max<long long>(1, std::min<long long>(threads, ((long long)m + (1 << 16) - 1) >> 16));
    std::vector<std::array<int, 7>> ccnt((size_t)cparts + 1);
    for (auto &c : ccnt) c.fill(0);
    parallel_for(cparts, cparts, 1, [&](long long t0, long long t1) {
        for (long long t = t0; t < t1; ++t) {
            std::array<int, 7> c{};
            c.fill(0);
            for (long long i = (long long)m * t / cparts, e = (long long)m * (t + 1) / cparts; i < e; ++i) { const int r = scan ? scan[i] : (int)i; c[(size_t)cat_of(rp[r + 1] - rp[r])]++; }
            ccnt[(size_t)t + 1] = c;
        }
    });
    for (int t = 0; t < cparts; ++t) for (int k = 0; k < 7; ++k) ccnt[(size_t)t + 1][(size_t)k] += ccnt[(size_t)t][(size_t)k];     // -> first index of part t
    n1 = ccnt[(size_t)cparts][0]; n3 = ccnt[(size_t)cparts][1]; n2 = ccnt[(size_t)cparts][2]; nz0 = ccnt[(size_t)cparts][3];
    n4 = ccnt[(size_t)cparts][4]; nlong = ccnt[(size_t)cparts][5]; nmed = ccnt[(size_t)cparts][6];
    std::vector<int> rid1(n1), rid2(n2), rid3(n3), rid4(n4), rid0(nz0), ridL(nlong);
    raw_vector<int> ridM_in((size_t)nmed);       // every element is written by the fill below; no 33 MB zero fill for 8 M rows
    {
        int *const lists[7] = {rid1.data(), rid3.data(), rid2.data(), rid0.data(), rid4.data(), ridL.data(), ridM_in.data()};
        parallel_for(cparts, cparts, 1, [&](long long t0, long long t1) {
            for (long long t = t0; t < t1; ++t) {
                std::array<int, 7> at = ccnt[(size_t)t];
                for (long long i = (long long)m * t / cparts, e = (long long)m * (t + 1) / cparts; i < e; ++i) {
                    const int r = scan ? scan[i] : (int)i;
                    const int k = cat_of(rp[r + 1] - rp[r]);
                    lists[k][at[(size_t)k]++] = r;
                }
            }
        });
    }
    const int n1_all = n1, n3_all = n3;
    const int nnz_short = n1 + 3 * n3 + 2 * n2 + 4 * n4;
    long long nnz_long = 0;
    for (int r : ridL) nnz_long += rp[r + 1] - rp[r];

    // 1&3 pairing count: dasp_f64.h:597-607 (multiple of 8) / dasp_f16.h:1127-1137 (multiple of 32).
    // Kept only because it fixes which slots of the output permutation those rows occupy.
    int c13 = std::min(n1, n3);
    if (c13 / 8 >= 16) { c13 = f16 ? 32 * (c13 / 32) : 8 * (c13 / 8); n1 -= c13; n3 -= c13; }
    else c13 = 0;

    lap("classify");
    // ---- medium rows sorted by length, descending and stable (what utils.h:118-160,196-203 produce);
    // lengths are < block_longest, so one counting pass does it.
    raw_vector<int> ridM((size_t)nmed), lenM((size_t)nmed);
    {
        // parallel counting sort: per-part histograms, then every (length, part) pair gets its start -- lengths descending, parts in
        // row order inside a length: stable
        const int nbk = std::max(block_longest, 6) + 1;
        const int sparts = (int)std::max<long long>(1, std::min<long long>(threads, ((long long)nmed + (1 << 16) - 1) >> 16));
        std::vector<int> hist((size_t)sparts * (size_t)nbk, 0);
        parallel_for(sparts, sparts, 1, [&](long long t0, long long t1) {
            for (long long t = t0; t < t1; ++t) {
                int *h = hist.data() + (size_t)t * (size_t)nbk;
                for (long long i = (long long)nmed * t / sparts, e = (long long)nmed * (t + 1) / sparts; i < e; ++i) { const int r = ridM_in[(size_t)i]; h[rp[r + 1] - rp[r]]++; }
            }
        });
        int run = 0;
        for (int L = nbk - 1; L >= 0; --L)
            for (int t = 0; t < sparts; ++t) { int &c = hist[(size_t)t * (size_t)nbk + (size_t)L]; const int n = c; c = run; run += n; }
        parallel_for(sparts, sparts, 1, [&](long long t0, long long t1) {
            for (long long t = t0; t < t1; ++t) {
                int *h = hist.data() + (size_t)t * (size_t)nbk;
                for (long long i = (long long)nmed * t / sparts, e = (long long)nmed * (t + 1) / sparts; i < e; ++i) {
                    const int r = ridM_in[(size_t)i], L = rp[r + 1] - rp[r], at = h[L]++;
                    ridM[(size_t)at] = r; lenM[(size_t)at] = L;
                }
            }
        });
    }

    lap("sort medium");
    RefGeometry refg;
    {
        std::vector<int> lenL(ridL.size());
        for (size_t i = 0; i < ridL.size(); ++i) lenL[i] = rp[ridL[i] + 1] - rp[ridL[i]];
        refg = reference_geometry(f16, m, p.n, nnz, n1, c13, n3, n4, n2, lenL, lenM, threshold, threads);
    }
    lap("reference geometry");
    // ---- output permutation (order_rid): dasp_f64.h:960-976 / dasp_f16.h:1253-1270
    const int base_s = nlong + nmed;   // (all medium rows: the slab split happens below)
    const int pg = f16 ? 32 : 8;
    {
        auto linear = [](int count, int base) { SlotMap s{}; s.split = count; s.base[0] = base; s.base[1] = base; return s; };
        SlotMap m1{}, m3{};
        int b1, b13, b3, b4, b2, b0;
        if (!f16) { b1 = base_s; b13 = b1 + n1; b3 = b13 + 2 * c13; b4 = b3 + n3; b2 = b4 + n4; b0 = b2 + n2; }
        else { b13 = base_s; b3 = b13 + 2 * c13; b4 = b3 + n3; b2 = b4 + n4; b1 = b2 + n2; b0 = b1 + n1; }
        // len-1 list = rows in row order: first n1 unpaired, last c13 paired
        m1.split = n1; m1.base[0] = b1; m1.grp[0] = 0; m1.off[0] = 0;
        m1.base[1] = b13; m1.grp[1] = pg; m1.off[1] = 0;
        // len-3 list: first c13 paired, rest unpaired
        m3.split = c13; m3.base[0] = b13; m3.grp[0] = pg; m3.off[0] = pg;
        m3.base[1] = b3; m3.grp[1] = 0; m3.off[1] = 0;
        p.grp[0].len = 1; p.grp[0].count = n1_all; p.grp[0].map = m1;
        p.grp[1].len = 2; p.grp[1].count = n2; p.grp[1].map = linear(n2, b2);
        p.grp[2].len = 3; p.grp[2].count = n3_all; p.grp[2].map = m3;
        p.grp[3].len = 4; p.grp[3].count = n4; p.grp[3].map = linear(n4, b4);
        p.grp[4].len = 0; p.grp[4].count = nz0; p.grp[4].map = linear(nz0, b0);
        for (int g = 5; g < kNumShortGroups; ++g) { p.grp[g] = ShortGroup{}; p.grp[g].len = g; }
    }
    // ---- medium rows short enough to be stored as uniform-length slabs instead of MFMA blocks (opt.slab_max_len): the
    // sorted order puts them last, each length a contiguous run, so a slab's rows keep the slots nlong + position.
    // Defaults from tools/rowlen_probe.py (8 M rows of one length, fraction of the roofline as blocks / as slabs).
    int slab_max = std::min(kSlabMaxLen, std::max(4, p.opt.slab_max_len > 0 ? p.opt.slab_max_len : (f16 ? kSlabDefaultF16 : kSlabDefaultF64)));
    const int nmed_all = nmed;
    int nmf = nmed;
    while (nmf > 0 && lenM[nmf - 1] <= slab_max) --nmf;               // lenM is sorted descending
    if (p.opt.slab_max_len <= 0 && nmf < nmed_all && !meta_only) {
        // auto: a slab lane owns whole rows, so the 64 lanes of a step gather x for 128 / 256 different rows.  That only beats
        // the blocks when neighbouring rows read neighbouring columns (stencils: the step's gathers coalesce, 8 M rows of 5:
        // 0.39 -> 0.82 of the roofline); on graph-like rows every lane hits another line and fewer, longer waves lose
        // (webbase-1M 16.9 -> 22 us, ljournal-2008 0.61 -> 0.70 ms, cop20k_A loses its LDS windows: 11 -> 32 us).
        // Measure it: pairs of successive equally long candidate rows, share of positions whose columns differ by < 16.
        std::vector<int> pairs;
        const int cand = nmed_all - nmf, step = std::max(2, (cand / 4096) & ~1);
        for (int i = nmf; i + 1 < nmed_all; i += step)
            if (lenM[i] == lenM[i + 1]) { pairs.push_back(ridM[i]); pairs.push_back(ridM[i + 1]); }
        long long near = 0, entries = 0;
        // "neighbouring": within a 128-byte line of x for f64.  f16 (late r5): within 512 columns -- an f16 tile holds 16 columns of a row, so rows of fewer than 12 have NO regular
        // chunk at all (a 16 x 16 tile is never 75 % full) and their MFMA blocks are all tail steps: 8 M local rows of 5..8 run at 0.21 of the roofline as blocks, 0.62 as slabs; a
        // circuit-like mix 0.32 -> 0.60; rows of 17 0.41 -> 0.61 (tools/scratch/circ_probe.py, tools/category_knobs.py).  The graph stand-ins, whose equally long neighbours are far
        // apart in the matrix, stay blocks (slabs cost webbase-1M f16 15.1 -> 18.8 us, rmat_2M 0.134 -> 0.164 ms: tools/r5_f16_slab_probe.sh)
        // Only where those degenerate rows are most of the candidates: rows of 12..24 in a band keep their blocks and with them the LDS windows (4.2)
        long long cand_nnz = 0, tiny_nnz = 0;
        for (int i = nmf; i < nmed_all; ++i) { cand_nnz += lenM[i]; if (lenM[i] < 12) tiny_nnz += lenM[i]; }
        // r6: ... or where ONE length holds >= 80 % of the candidates' nonzeros (13- / 19-point stencils, fixed-valence meshes in f16): rows of one length keep their
        // neighbours under the length sort, and as slabs they run half as long again (r5, profiles/r05_category_sweep.md 16: 4 M rows of 17 0.413 -> 0.615 of the roofline,
        // 3 M rows of 24 0.445 -> 0.666), while thirteen lengths 12..24 -- whose equally long rows lie 13 rows apart -- keep the windows (0.568 against 0.484)
        long long dom_nnz = 0;
        for (int i = nmf; i < nmed_all; ) { int j = i; while (j < nmed_all && lenM[j] == lenM[i]) ++j; dom_nnz = std::max(dom_nnz, (long long)(j - i) * lenM[i]); i = j; }
        // ... whose OWN columns run along lines of x (a slab lane walks its row: consecutive steps then hit the line the last one fetched): under 30 % of a sampled row's
        // entries on another 128-byte line than their predecessor.  Rows of 12 with columns anywhere within +-500 of the diagonal pass the neighbour test too and LOSE as slabs
        // (120 k rows, f16: 6.3 us against 4.2 as blocks, gpurun_out/r6/call21.txt)
        bool dom_runs = false;
        if (f16 && 5 * dom_nnz >= 4 * cand_nnz && !pairs.empty()) {
            long long lines = 0, ent = 0;
            if (dev) { std::vector<int> firsts; for (size_t q = 0; q < pairs.size(); q += 2) firsts.push_back(pairs[q]); if (int rc = devpack_line_scatter(p, *dev, firsts, &lines, &ent)) return rc; }
            else
            for (size_t q = 0; q < pairs.size(); q += 2) {
                const int b = rp[pairs[q]], e = rp[pairs[q] + 1];
                int prev = remap(ci[b]) >> 6; lines += 1;
                for (int j = b + 1; j < e; ++j) { const int l = remap(ci[j]) >> 6; lines += l != prev; prev = l; }
                ent += e - b;
            }
            dom_runs = ent > 0 && 10 * lines < 3 * ent;
        }
        const int within = f16 && (2 * tiny_nnz >= cand_nnz || dom_runs) ? 512 : 16;
        if (dev) { if (int rc = devpack_row_coherence(p, *dev, pairs, within, &near, &entries)) return rc; }
        else
            for (size_t q = 0; q + 1 < pairs.size(); q += 2) {
                const int a = rp[pairs[q]], b = rp[pairs[q + 1]], len = rp[pairs[q] + 1] - a;
                for (int k = 0; k < len; ++k) { const int d = remap(ci[a + k]) - remap(ci[b + k]); near += d > -within && d < within; }
                entries += len;
            }
        // ... and only when those rows are most of the matrix: a minority of slab waves (128 rows x up to 16 sequential steps) next
        // to MFMA blocks only lengthens the tail (nlpkkt160's boundary rows: x0.01 5.7 -> 7.9 us, x0.03 15.4 -> 17.4 us)
        long long slab_nnz = 0;
        for (int i = nmf; i < nmed_all; ++i) slab_nnz += lenM[i];
        if (entries == 0 || (double)near < 0.5 * (double)entries || 2 * slab_nnz < (long long)nnz) { slab_max = 4; nmf = nmed_all; }
    }
    std::vector<std::vector<int>> slab_rows((size_t)kSlabMaxLen + 1);
    for (int i = nmf; i < nmed_all; ) {
        const int L = lenM[i];
        int j = i;
        while (j < nmed_all && lenM[j] == L) ++j;
        slab_rows[L].assign(ridM.begin() + i, ridM.begin() + j);
        ShortGroup &G = p.grp[L];                                       // group index == row length for L >= 5
        G.count = j - i; G.map = SlotMap{}; G.map.split = G.count; G.map.base[0] = G.map.base[1] = nlong + i;
        i = j;
    }
    const std::vector<int> *glist[kNumShortGroups] = {&rid1, &rid2, &rid3, &rid4, &rid0};
    for (int g = 5; g < kNumShortGroups; ++g) glist[g] = &slab_rows[g];
    p.order.assign((size_t)m, -1);
    for (int i = 0; i < nlong; ++i) p.order[i] = ridL[i];
    parallel_for(nmed, threads, 1 << 16, [&](long long b, long long e) { for (long long i = b; i < e; ++i) p.order[(size_t)nlong + (size_t)i] = ridM[(size_t)i]; });
    for (int g = 0; g < kNumShortGroups; ++g) {
        const ShortGroup &G = p.grp[g];
        const std::vector<int> &list = *glist[g];
        parallel_for(G.count, threads, 1 << 16, [&](long long b, long long e) { for (long long t = b; t < e; ++t) p.order[(size_t)G.map.slot((int)t)] = list[(size_t)t]; });
    }
    // from here on "medium" means the MFMA part only
    if (nmf < nmed_all) {
        ridM.resize((size_t)nmf); lenM.resize((size_t)nmf);
        raw_vector<int> keep; keep.reserve((size_t)nmf);
        for (int r : ridM_in) if (rp[r + 1] - rp[r] > slab_max) keep.push_back(r);
        ridM_in.swap(keep);
        nmed = nmf;
    }
    const int nlong_cls = nlong;                      // the reference's row_long
    p.n_mfma_rows = nmf;
    const bool natural = p.opt.y_order == DASP_Y_NATURAL;
    const bool mapped = natural && !p.dst_map.empty();     // a panel writing into its parent's slot order
    auto rowdst = [&](int row) { return mapped ? p.dst_map[row] : row; };
    auto ydst = [&](int slot) { return natural ? rowdst(p.order[slot]) : slot; };

    lap("order_rid");
    // ---- optional windowed order for the medium rows (LDS-staged x).  The reference sorts all
    // medium rows globally by length, which scatters the 16 rows of a block over the matrix; here rows are sorted inside
    // windows of `row_window` consecutive medium rows only, one window per workgroup, so a workgroup's rows share a narrow
    // span of x that is staged once in LDS.  Output slots stay the reference's (order_rid untouched): y goes through med_dst.
    p.windowed = false; p.win_hybrid = false; p.win_rel16 = false; p.row_window = 0; p.lds_bytes = 0;
    p.med_dst.clear(); p.win_cmin.clear(); p.win_len.clear();
    double window_frac = 0.0;
    if ((p.opt.x_window >= 0 || p.opt.x_window == -2) && nmed > 0 && !meta_only) {
        // default window height: every window pays the same x copy whatever its height, so as few windows as still occupy the
        // 256 CUs -- one workgroup each, or two each (2 x 80 KiB of LDS) once a window would pass 1024 rows.  Just under the
        // CU count, not at it: sweeps on the cop20k_A stand-in put the optimum at 200-230 windows (162 k medium rows: 318 / 255 /
        // 231 / 212 / 159 windows = 17.7 / 18.6 / 14.6 / 14.6 / 16.0 us; 130 k: 255 / 226 / 170 = 15.4 / 13.2 / 13.4; 65 k:
        // 507 / 254 / 127 / 64 = 8.4 / 7.4 / 9.1 / 13.4) and at ~424 for 325 k rows (424 / 318 = 24.6 / 27.3 us).
        // r6 (profiles/r06_small_matrix.md): with the 128-register build exactly one window workgroup resides per CU, so the fastest grid is the one that puts a
        // workgroup on EVERY CU and none behind another: windows (rounded up to a whole number per XCD, as the kernel deals them) + long-piece workgroups + short-tile
        // workgroups (none where the tiles fold into the windows) <= 256.  cop20k_A: 212 windows of 512 rows 10.56 us, 242 of 448 10.08, 251 of 432 WITH seven short-tile
        // workgroups behind them 17.6 (263 workgroups: seven CUs take a second one) -- hence the count of ALL workgroups, and window heights in whole blocks (16 rows).
        int R = p.opt.row_window;
        if (R <= 0) {
            const int cus = 256, SR = geo.short_rows;
            int est_tiles = 0;
            for (int g = 0; g < kNumShortGroups; ++g) est_tiles += ceil_div(p.grp[g].count, SR);
            long long est_pieces = nlong;
            for (int i = 0; i < nlong; ++i) est_pieces += (rp[ridL[(size_t)i] + 1] - rp[ridL[(size_t)i]]) / 1024;
            for (int r = 128; r <= 1024 && R <= 0; r += kMedRows) {
                const int nWr = ceil_div(nmed, r), wpw_r = std::min(16, r / kMedRows);
                const int wgs = ceil_div(nWr, 8) * 8 + (int)ceil_div(est_pieces, (long long)wpw_r) + (win_fold_tiles(nWr, est_tiles) ? 0 : ceil_div(est_tiles, wpw_r));
                if (wgs <= cus) R = r;
            }
            // more rows than 256 windows of 1024 hold: two workgroups per CU (the 64-register build, 2 x 80 KiB of LDS), ~448 windows as before
            if (R <= 0) R = std::max(128, ceil_div(ceil_div(nmed, 448), 64) * 64);
        }
        R = std::min(1024, std::max(64, (R / kMedRows) * kMedRows));         // whole blocks; <= 16 waves per workgroup, 1-4 blocks per wave
        // default cap: 80 KiB = two workgroups per CU out of gfx950's 160 KiB of LDS
        const bool order_only = p.opt.x_window == -2;          // windowed order, no LDS staging (every window gathers from global memory)
        int cap_bytes = order_only ? 0 : (p.opt.x_window > 0 ? std::min(p.opt.x_window, kWinLdsMax) : 80 * 1024);
        const int A = 16 / geo.vbytes;                         // window base aligned for 16-byte copies
        // medium rows in row order, then a stable descending length sort inside each window
        raw_vector<int> ridW((size_t)nmed), lenW((size_t)nmed);
        {
            const raw_vector<int> &rows = ridM_in;             // the medium rows in row order
            const int nW = ceil_div(nmed, R);
            parallel_for(nW, threads, 8, [&](long long w0, long long w1) {
                std::vector<int> bucket;
                for (long long w = w0; w < w1; ++w) {
                    const int a0 = (int)w * R, a1 = std::min(nmed, a0 + R);
                    bucket.assign((size_t)block_longest + 1, 0);
                    for (int i = a0; i < a1; ++i) bucket[rp[rows[i] + 1] - rp[rows[i]]]++;
                    int run = a0;
                    for (int L = block_longest; L >= 0; --L) { int c = bucket[L]; bucket[L] = run; run += c; }
                    for (int i = a0; i < a1; ++i) { const int L = rp[rows[i] + 1] - rp[rows[i]]; const int at = bucket[L]++; ridW[at] = rows[i]; lenW[at] = L; }
                }
            });
        }
        const int nW = ceil_div(nmed, R);
        std::vector<int> cmin(nW), wlen(nW);
        std::vector<long long> wnnz(nW);
        std::vector<int> wlo(nW), whi(nW);
        // span [lo, hi] and nonzeros of the windows over rows list[0 .. n): one window per R rows
        auto spans_of = [&](const raw_vector<int> &list, int n, int *lo_out, int *hi_out, long long *nnz_out) -> int {
            if (dev) return devpack_window_spans(p, *dev, list, R, lo_out, hi_out, nnz_out);
            parallel_for(ceil_div(n, R), threads, 8, [&](long long w0, long long w1) {
                for (long long w = w0; w < w1; ++w) {
                    const int a0 = (int)w * R, a1 = std::min(n, a0 + R);
                    int lo = 2147483647, hi = -1; long long k = 0;
                    for (int i = a0; i < a1; ++i) {
                        const int r = list[(size_t)i];
                        for (int j = rp[r]; j < rp[r + 1]; ++j) { const int c = remap(ci[j]); lo = std::min(lo, c); hi = std::max(hi, c); }
                        k += rp[r + 1] - rp[r];
                    }
                    lo_out[w] = lo; hi_out[w] = hi; nnz_out[w] = k;
                }
            });
            return DASP_OK;
        };
        // auto mode on a large matrix: look at 64 evenly spaced windows first.  If under a quarter of THEIR nonzeros sit in windows
        // that would fit even the whole 160 KiB of LDS, no strict window rule can reach its 50 % -- skip reading every column id of the
        // matrix for nothing (HV15R: 332 ms of the host path's 659, 6 of the device path's 38)
        bool scan_all = true;
        if (p.opt.x_window == 0 && nW > 256) {
            const int ns = 64;
            raw_vector<int> sample; sample.reserve((size_t)ns * (size_t)R);
            for (int q = 0; q < ns; ++q) {
                const int w = (int)((long long)q * (nW - 1) / ns);                 // never the (possibly partial) last window
                sample.insert(sample.end(), ridW.begin() + (size_t)w * R, ridW.begin() + (size_t)(w + 1) * R);
            }
            std::vector<int> slo(ns), shi(ns);
            std::vector<long long> snz(ns);
            if (int rc = spans_of(sample, ns * R, slo.data(), shi.data(), snz.data())) return rc;
            long long fit_s = 0, all_s = 0;
            for (int q = 0; q < ns; ++q) {
                all_s += snz[q];
                if (shi[q] >= 0 && ((long long)shi[q] - (slo[q] / A) * A + 1) * geo.vbytes <= kWinLdsMax) fit_s += snz[q];
            }
            scan_all = 4 * fit_s >= all_s;
        }
        if (scan_all) { if (int rc = spans_of(ridW, nmed, wlo.data(), whi.data(), wnnz.data())) return rc; }
        else
            parallel_for(nW, threads, 64, [&](long long w0, long long w1) {
                for (long long w = w0; w < w1; ++w) {                               // "does not fit": what the full scan would have concluded
                    wlo[(size_t)w] = 2147483647; whi[(size_t)w] = -1;
                    long long k = 0;
                    for (long long i = w * R, e = std::min<long long>(nmed, (w + 1) * R); i < e; ++i) k += lenW[(size_t)i];
                    wnnz[(size_t)w] = k;
                }
            });
        long long fit = 0, all = 0; int maxlen = 0;
        auto fit_windows = [&]() {
            fit = 0; all = 0; maxlen = 0;
            for (int w = 0; w < nW; ++w) {
                const int lo = (wlo[w] / A) * A, hi = whi[w];
                const long long span = (long long)hi - lo + 1;
                if (hi >= 0 && span * geo.vbytes <= cap_bytes) { cmin[w] = lo; wlen[w] = (int)span; }
                else { cmin[w] = 0; wlen[w] = 0; }
                all += wnnz[w];
                if (wlen[w] > 0) { fit += wnnz[w]; maxlen = std::max(maxlen, wlen[w]); }
            }
            window_frac = all > 0 ? (double)fit / (double)all : 0.0;
        };
        fit_windows();
        // auto: when the spans are too wide for two workgroups per CU (80 KiB each), one workgroup per CU with all of the LDS
        // still beats global gathers on band-scattered rows (+-30 k columns, f16: 98.8 -> 65.9 us; +-8 k, f64: 166.8 -> 115.8 us)
        if (p.opt.x_window == 0 && window_frac < 0.5) { cap_bytes = kWinLdsMax; fit_windows(); }
        const bool force = p.opt.x_window > 0;
        // ---- hybrid windows: when the whole span of a window does not fit (graph-like rows: most columns near the rows -- the
        // pages of a host, the members of a community -- plus a scattered remainder), stage the DENSEST span of cap bytes and let
        // the gathers that fall outside it read global memory (kernel: XHyb).  Host CSR only.  It pays on rows of even length whose
        // columns form a band plus outliers (tools/hybrid_probe.py: 2 M rows of 14, +-3000 band + 10 % anywhere, f64 0.234 -> 0.189 ms,
        // f16 0.086 -> 0.061 ms; +-8000 + 3 %: 0.190 -> 0.126 ms); on the power-law stand-ins the window workgroups themselves cost
        // more than the LDS gathers save (webbase-1M 14.6 -> 22.4 us with 85 % of the medium gathers staged, ljournal-2008
        // 0.545 -> 0.587 ms with 53 %: one 1024-thread workgroup per window balances rows of 5..255 nonzeros badly).  So auto
        // asks for even rows (longest <= 4 x the mean) on top of: strict windows cover < half, densest spans cover >= 60 %.
        p.win_hybrid = false;
        const bool even_rows = nmed > 0 && (long long)lenM[0] * nmed <= 4 * std::max<long long>(1, all);
        if (!dev && !order_only && p.opt.x_window_hybrid >= 0 &&
            (p.opt.x_window_hybrid > 0 || (p.opt.x_window == 0 && window_frac < 0.5 && even_rows))) {
            const int hcap = p.opt.x_window > 0 ? std::min(p.opt.x_window, kWinLdsMax) : 80 * 1024;    // two workgroups per CU
            const long long cap_cols = std::max<long long>(A, (hcap / geo.vbytes / A) * A);
            const long long xl = p.opt.n_parts > 0 ? (long long)p.opt.n_parts * p.opt.part_stride : (long long)p.n;
            std::vector<int> hmin(nW), hlen(nW);
            std::vector<long long> hin(nW);
            const bool force_sort = std::getenv("DASP_HYBRID_SORT") != nullptr;       // test knob: the sort-based search (tests compare the two)
            // auto mode on a large matrix: the densest spans of 64 evenly spaced windows first -- when they hold under 55 % of those
            // windows' gathers the whole will not reach 60 %, and sorting every window's columns is skipped (HV15R: 0.33 s of the host
            // path, its three-plane rows give 33 %; nlpkkt160: 0.45 s, ~50 %)
            std::vector<int> wlist;
            if (p.opt.x_window_hybrid == 0 && nW > 256) for (int q = 0; q < 64; ++q) wlist.push_back((int)((long long)q * (nW - 1) / 64));
            bool sampled = !wlist.empty();
            for (int pass = sampled ? 0 : 1; pass < 2; ++pass) {
            if (pass == 1) { wlist.resize((size_t)nW); for (int w = 0; w < nW; ++w) wlist[(size_t)w] = w; }
            parallel_for((long long)wlist.size(), threads, 4, [&](long long w0, long long w1) {
                std::vector<int> cols, hist;
                for (long long wi = w0; wi < w1; ++wi) {
                    const int w = wlist[(size_t)wi];
                    const int a0 = (int)w * R, a1 = std::min(nmed, a0 + R);
                    cols.clear();
                    int cmin_w = 2147483647, cmax_w = -1;
                    for (int i = a0; i < a1; ++i) {
                        const int r = ridW[i];
                        for (int j = rp[r]; j < rp[r + 1]; ++j) { const int c = remap(ci[j]); cols.push_back(c); cmin_w = std::min(cmin_w, c); cmax_w = std::max(cmax_w, c); }
                    }
                    // best 16-byte-aligned start: for every aligned start that holds an entry, the entries in [start, start + cap_cols); the
                    // first start with the largest count wins.  (An entry c that is not the smallest of its aligned group counts fewer entries
                    // than the group's smallest one -- which is itself inside [group start, c) -- so only group starts compete.)
                    long long best = 0; int bs = 0;
                    const long long g0 = cols.empty() ? 0 : cmin_w / A, ngrp = cols.empty() ? 0 : cmax_w / A - g0 + 1;
                    if (!cols.empty() && ngrp <= 4 * (long long)cols.size() + 1024 && !force_sort) {
                        // dense enough: a histogram over the aligned groups and a sliding sum -- O(entries + groups) instead of a sort
                        hist.assign((size_t)ngrp + 1, 0);
                        for (int c : cols) hist[(size_t)(c / A - g0)]++;
                        const long long wg = cap_cols / A;                                  // groups per span
                        long long run = 0;
                        for (long long g = 0; g < std::min(wg, ngrp); ++g) run += hist[(size_t)g];
                        for (long long g = 0; g < ngrp; ++g) {
                            if (hist[(size_t)g] > 0 && run > best) { best = run; bs = (int)((g0 + g) * A); }
                            run -= hist[(size_t)g];
                            if (g + wg < ngrp) run += hist[(size_t)(g + wg)];
                        }
                    } else {
                    std::sort(cols.begin(), cols.end());
                    size_t hi_i = 0;
                    for (size_t lo_i = 0; lo_i < cols.size(); ++lo_i) {
                        if (lo_i && cols[lo_i] == cols[lo_i - 1]) continue;
                        const long long st = (cols[lo_i] / A) * A;
                        if (hi_i < lo_i) hi_i = lo_i;
                        while (hi_i < cols.size() && cols[hi_i] < st + cap_cols) ++hi_i;
                        // entries between the aligned start and cols[lo_i] belong to the span too, but they are < A away: ignore them
                        const long long cnt = (long long)(hi_i - lo_i);
                        if (cnt > best) { best = cnt; bs = (int)st; }
                    }
                    }
                    hmin[w] = bs; hin[w] = best;
                    hlen[w] = (int)std::min<long long>(cap_cols, xl - bs);
                }
            });
            if (pass == 0) {
                long long in_s = 0, all_s = 0;
                for (int w : wlist) { in_s += hin[(size_t)w]; all_s += wnnz[(size_t)w]; }
                if (100 * in_s < 55 * all_s) { std::fill(hin.begin(), hin.end(), 0); break; }     // cover stays 0: no hybrid windows
            }
            }
            long long in = 0;
            for (int w = 0; w < nW; ++w) in += hin[w];
            const double cover = all > 0 ? (double)in / (double)all : 0.0;
            if (p.opt.x_window_hybrid > 0 || cover >= 0.6) {
                p.win_hybrid = true;
                fit = 0; maxlen = 0;
                for (int w = 0; w < nW; ++w) {
                    // a window whose best span holds under a quarter of its gathers is not worth its copy: global gathers only
                    if (hin[w] * 4 < wnnz[w] || hlen[w] <= 0) { cmin[w] = 0; wlen[w] = 0; continue; }
                    cmin[w] = hmin[w]; wlen[w] = hlen[w]; fit += hin[w]; maxlen = std::max(maxlen, wlen[w]);
                }
                window_frac = all > 0 ? (double)fit / (double)all : 0.0;
            }
        }
        bool worth = window_frac >= 0.5 && (double)all >= 0.5 * (double)nnz;
        if (worth && !force && !order_only) {
            // staging pays only when a row's gathers are scattered over the window: if neighbouring nonzeros of a row share
            // 128-byte lines of x anyway (FEM / stencil rows: runs of adjacent columns), the L1 already serves them and the
            // copy + window workgroups are pure overhead (HV15R x0.03: 18.2 -> 24.5 us, nlpkkt160 x0.01: 5.7 -> 8.5 us with
            // windows; line-scatter 0.08-0.37 on the FEM-like stand-ins, 0.89-0.97 on the cop20k_A band)
            std::vector<int> sample;
            const int step = std::max(1, nmed / 2048);
            for (int i = 0; i < nmed; i += step) sample.push_back(ridM_in[i]);
            long long lines = 0, entries = 0;
            if (dev) { if (int rc = devpack_line_scatter(p, *dev, sample, &lines, &entries)) return rc; }
            else {
                const int shift = geo.vbytes == 8 ? 4 : 6;
                for (int r : sample) {
                    const int b = rp[r], e = std::min(rp[r + 1], b + 512);
                    int prev = remap(ci[b]) >> shift; lines += 1;
                    for (int j = b + 1; j < e; ++j) { const int l = remap(ci[j]) >> shift; lines += l != prev; prev = l; }
                    entries += e - b;
                }
            }
            worth = entries > 0 && (double)lines >= 0.6 * (double)entries;
            // r6: ... and only when a row's columns spread over more x than the L1 holds anyway.  The CORE span of a row -- 10th to 90th percentile of its columns, so that
            // a few outliers do not count -- times the value size, median over the sampled rows: under 16 KB the 16 rows of a block gather from a few KB each and hit the
            // L1, and the window's copy -> barrier -> blocks chain only costs (rows of 12 within +-500 columns + 10 % anywhere: 120 k rows f64 7.5 us with hybrid windows
            // against 5.3-5.7 without, f16 6.0 / 4.2; 1 M rows f64 61.5 / 44.2; 500 k rows of 14 within +-3000 f16 25.2 / 19.6 -- while +-8000 f64 (102 KB) 181 -> 151 us,
            // 200 k rows of 30 within +-2000 f64 (26 KB) 25.9 -> 22.7 and cop20k_A (51 KB) keep them: tools/hybrid_probe.py, gpurun_out/r6/call19.txt, call20.txt)
            if (worth && p.opt.x_window_hybrid <= 0) {      // (a caller who forces the hybrid windows keeps them)
                std::vector<int> fetched;
                if (dev) {
                    std::vector<long long> idx;
                    for (int r : sample) for (int j = rp[r], e = std::min(rp[r + 1], rp[r] + 512); j < e; ++j) idx.push_back(j);
                    if (int rc = devpack_gather_columns(p, *dev, &idx, 0, 0, 0, fetched)) return rc;
                }
                std::vector<long long> spans;
                std::vector<int> cols;
                size_t fpos = 0;
                for (int r : sample) {
                    const int b = rp[r], e = std::min(rp[r + 1], b + 512), len = e - b;
                    cols.resize((size_t)len);
                    if (dev) { for (int j = 0; j < len; ++j) cols[(size_t)j] = fetched[fpos + (size_t)j]; fpos += (size_t)len; }
                    else for (int j = 0; j < len; ++j) cols[(size_t)j] = remap(ci[b + j]);
                    if (len < 4) continue;
                    std::sort(cols.begin(), cols.end());
                    spans.push_back((long long)cols[(size_t)(len - 1 - len / 10)] - cols[(size_t)(len / 10)] + 1);
                }
                // ... provided the global length sort leaves neighbouring rows in one block (rows of one length: the block's 16 rows then share those few KB).  Rows of many
                // lengths end up in blocks of rows from all over the matrix -- 16 such regions per chunk -- and keep their windows whatever their own span: cop20k_A in f16
                // (12.8 KB per row) 6.9 -> 6.5 us, x4 22.0 -> 17.3 with windows.  The measure of the rule below: sampled blocks of the sorted order whose rows lie far apart.
                int far = 0, cnt = 0;
                {
                    const int nblk = nmed / kMedRows, bstep = std::max(1, nblk / 512);
                    for (int b = 0; b < nblk; b += bstep) {
                        int lo = 2147483647, hi = -1;
                        for (int i = b * kMedRows; i < (b + 1) * kMedRows; ++i) { lo = std::min(lo, ridM[i]); hi = std::max(hi, ridM[i]); }
                        far += hi - lo > 16 * kMedRows; ++cnt;
                    }
                }
                if (!spans.empty() && 2 * far < cnt) {
                    std::nth_element(spans.begin(), spans.begin() + (long long)(spans.size() / 2), spans.end());
                    if (spans[spans.size() / 2] * geo.vbytes < 16 * 1024) worth = false;
                }
            }
            // ... or when it is the GLOBAL LENGTH SORT that scatters: rows of many different lengths whose columns are local (the strict windows fit, so they are) end up in
            // blocks of 16 rows from all over the matrix -- 16 regions of x per chunk whatever the rows' own runs.  Windows sort inside R rows only and stage those rows' x
            // (late r5, tools/scratch/window_order_probe.py / window_size_probe.py: 1 M local rows of 5..255 f64 0.571 -> 0.792 of the roofline, f16 0.623 -> 0.892; 4 M rows of
            // 10..60 0.628 -> 0.722 / 0.546 -> 0.780; pays from ~8 M nonzeros on, so: >= 16 M).  Rows of one length (HV15R, Queen_4147) keep their row order in the sort and are
            // not touched: measured as the span of row ids inside sampled blocks of the sorted order
            if (!worth && nnz >= (16 << 20) && nmed >= 16 * 64) {
                int far = 0, cnt = 0;
                const int nblk = nmed / kMedRows, bstep = std::max(1, nblk / 512);
                for (int b = 0; b < nblk; b += bstep) {
                    int lo = 2147483647, hi = -1;
                    for (int i = b * kMedRows; i < (b + 1) * kMedRows; ++i) { lo = std::min(lo, ridM[i]); hi = std::max(hi, ridM[i]); }
                    // (not neighbours: more than 16 rows apart on average -- every row of the block then reads lines of its own; f16, whose 2-byte gathers use a 64th of every
                    // line they miss: more than 4 apart -- 6 M tetra-like local rows of ~N(15,4) nonzeros, neighbours ~10 apart: f16 0.404 -> 0.514 with windows, f64 0.625 -> 0.653)
                    far += hi - lo > (f16 ? 4 : 16) * kMedRows; ++cnt;
                }
                worth = cnt > 0 && 2 * far >= cnt;
            }
        }
        if (!(((force || worth) && fit > 0) || order_only)) p.win_hybrid = false;
        if (((force || worth) && fit > 0) || order_only) {
            p.windowed = true; p.row_window = R;
            p.lds_bytes = ((maxlen * geo.vbytes + 255) / 256) * 256;
            p.win_cmin.swap(cmin); p.win_len.swap(wlen);
            std::vector<int> slot_of_row((size_t)m, -1);
            for (int i = 0; i < nmed; ++i) slot_of_row[ridM[i]] = nlong + i;       // reference slot of each (MFMA) medium row: p.med_slot0 + i
            p.med_dst.resize(nmed);
            for (int i = 0; i < nmed; ++i) p.med_dst[i] = natural ? rowdst(ridW[i]) : slot_of_row[ridW[i]];
            ridM.swap(ridW); lenM.swap(lenW);                                      // the packers below follow the windowed order
        }
    }

    lap("window decision");
    // ---- the longest medium rows as wave-sized pieces (opt.piece_min_len).  A 16-row block of rows with L nonzeros is a serial
    // chain of ceil(L / K) MFMA steps in ONE wave, each batch of steps one memory latency; on a small matrix with a power-law
    // tail the kernel then lasts as long as its longest block (webbase-1M stand-in, f16: rows of 255 = 16 steps = 8 batches ~ 8 us
    // of a 14.6 us kernel).  Such rows are stored like the reference's long rows instead -- CSR order, one wave per piece, all 64
    // lanes on one row -- while their slots in order_rid stay the medium rows' (they are the FIRST medium slots, right behind the
    // long rows, so "index in the combined piece list == slot" holds as it does for the long rows).  Classifier counters unchanged.
    int nsp = 0;
    if (nmf > 0 && !meta_only && p.opt.piece_min_len >= 0 && !p.windowed) {   // window rows keep their LDS gathers (cop20k_A: 11.1 -> 14.7 us when split)
        int Tmin = p.opt.piece_min_len > 0 ? std::max(5, p.opt.piece_min_len) : 0;
        if (Tmin == 0) {
            // auto: only when the matrix is latency-bound (a couple of batches of the longest block already take as long as
            // streaming the whole matrix at ~5 TB/s) and the rows concerned are a tail (<= 15 % of the nonzeros)
            const double est_us = (double)nnz * (geo.vbytes + 4) / 5.0e6;
            const int shot = f16 ? 2 : 8, batch = f16 ? 2 : 4;
            auto batches = [&](int L) { const int st = ceil_div(L, geo.med_k); return st <= shot ? 1 : ceil_div(st, batch); };
            const int allowed = std::max(2, (int)(0.25 * est_us / 0.8));
            if (batches(lenM[0]) > allowed) {
                int Lmax = lenM[0];
                while (Lmax > 5 && batches(Lmax) > allowed) --Lmax;
                Tmin = Lmax + 1;
            }
        }
        if (Tmin > 0) {
            // rows from the front (the longest) while they are >= Tmin and, in auto mode, stay a tail of <= 15 % of the nonzeros;
            // never cut inside a run of equal lengths (the row-order list is filtered by length below)
            const bool forced = p.opt.piece_min_len > 0;
            int cnt = 0; long long k = 0;
            while (cnt < nmf && lenM[cnt] >= Tmin && (forced || (k + lenM[cnt]) * 100 <= 15ll * nnz)) { k += lenM[cnt]; ++cnt; }
            while (cnt > 0 && cnt < nmf && lenM[cnt] == lenM[cnt - 1]) --cnt;
            nsp = cnt;
        }
    }
    if (nsp > 0) {
        const int Tcut = lenM[nsp - 1];
        ridL.insert(ridL.end(), ridM.begin(), ridM.begin() + nsp);
        ridM.erase(ridM.begin(), ridM.begin() + nsp); lenM.erase(lenM.begin(), lenM.begin() + nsp);
        raw_vector<int> keep; keep.reserve(ridM_in.size());
        for (int r : ridM_in) if (rp[r + 1] - rp[r] < Tcut) keep.push_back(r);
        ridM_in.swap(keep);
        nmf -= nsp; nmed = nmf; nlong += nsp;         // from here on nlong counts the rows STORED as pieces
    }
    p.med_slot0 = nlong;
    p.n_mfma_rows = nmf;
    long long nnz_pieces = 0;
    for (int r : ridL) nnz_pieces += rp[r + 1] - rp[r];
    // ---- long rows: compact, padded to kLongAlign; one wave per piece
    std::vector<long long> startL;
    int piece = p.opt.long_piece > 0 ? p.opt.long_piece : 1024;
    if (p.opt.long_piece <= 0) {
        // one piece per row, so that the second launch (long_reduce) disappears, when that leaves no wave with a long tail of its own: few
        // long nonzeros in rows of <= 16384 (launch-bound matrices: the launch costs ~2.7 us per SpMV on the webbase-1M stand-in, 19.7 ->
        // 17.0 us), or no row beyond 4096 at all (r4: the four panels of ljournal-2008 each paid a ~5 us launch for ~540 rows of 1025..2469)
        int longest = 0;
        for (int r : ridL) longest = std::max(longest, rp[r + 1] - rp[r]);
        const int one_piece = ab_knobs().one_piece_max;
        // ... unless the longest row alone would outlast the rest of the launch AND the stage-2 launch it saves (~3 us): one wave walks it in batches of BATCH
        // chunks, ~0.35 us each, against ~4 us + the matrix at ~5 TB/s for everything else (r5, tools/size_sweep.py: powerlaw_1M x0.03 -- 1.7 M nonzeros, hub rows of 15 728 -- f64 22.8 ->
        // 10.1 us, f16 10.6 -> 7.7 us in pieces of 1024; webbase-1M's 4700-long rows stay whole: 6.4 us of chain in a 29 us launch, 3.2 in 14.6 for f16)
        const double chain_us = 0.35 * (double)longest / (double)(geo.chunk * (f16 ? kMedBatch16 : kMedBatch64));
        const double rest_us = 4.0 + 3.0 + (double)nnz * (double)(geo.vbytes + 4) / 5.0e6;
        if ((nnz_pieces <= 2000000 && longest <= 16384 && chain_us <= rest_us) || longest <= one_piece) piece = std::max(piece, longest);
    }
    piece = std::max(geo.chunk, ceil_div(piece, geo.chunk) * geo.chunk);
    {
        std::vector<long long> start((size_t)nlong + 1, 0);
        for (int i = 0; i < nlong; ++i) {
            const int len = rp[ridL[i] + 1] - rp[ridL[i]];
            start[i + 1] = start[i] + (long long)ceil_div(len, kLongAlign) * kLongAlign;
        }
        const long long total = start[nlong];
        if (total >= (1LL << 31)) { set_error("long-row segment exceeds 2^31 elements"); return DASP_ERR_ARG; }
        p.cnt_long = (size_t)total;
        startL = start;
        if (pack) {
        p.long_val.resize((size_t)total * sizeof(T));          // not zero-filled: rows + their pads are written below
        p.long_cid.resize((size_t)total);
        }
        p.piece_ptr.clear(); p.piece_dst.clear(); p.multi_ptr.assign(1, 0); p.multi_dst.clear();
        int n_partial = 0;
        for (int i = 0; i < nlong; ++i) {
            const long long lp = start[i + 1] - start[i];
            const int np = (int)((lp + piece - 1) / piece);
            for (int q = 0; q < np; ++q) {
                p.piece_ptr.push_back((int)(start[i] + (long long)q * piece));
                p.piece_dst.push_back(np == 1 ? ydst(i) : ~(n_partial++));
            }
            if (np > 1) { p.multi_ptr.push_back(n_partial); p.multi_dst.push_back(ydst(i)); }
        }
        p.piece_ptr.push_back((int)total);
        T *lv = reinterpret_cast<T *>(p.long_val.data());
        // element-parallel (not row-parallel): a power-law matrix keeps most of its long nonzeros in a handful of rows
        if (pack) parallel_for(total, threads, 1 << 16, [&](long long b, long long e) {
            long long i = std::upper_bound(start.begin(), start.end(), b) - start.begin() - 1;   // row holding element b
            for (; i < nlong && start[i] < e; ++i) {
                const int r = ridL[i], len = rp[r + 1] - rp[r];
                const long long s0 = std::max(b, start[i]), s1 = std::min(e, start[i + 1]);
                const long long real_end = std::min(s1, start[i] + len);
                const long long src = (long long)rp[r] - start[i];
                for (long long j = s0; j < real_end; ++j) { lv[j] = val[src + j]; p.long_cid[(size_t)j] = remap(ci[src + j]); }
                for (long long j = std::max(s0, real_end); j < s1; ++j) { lv[j] = (T)0; p.long_cid[(size_t)j] = -1; }   // pad to kLongAlign
            }
        });
        // ---- 16-bit ids (plan.hpp long_cid16): the chunk grid is the kernel's -- CH elements from the piece's first element on
        {
            const int CHL = geo.chunk, np_all = (int)p.piece_dst.size();
            p.piece_c16.assign((size_t)2 * (size_t)np_all, 0);
            long long nchunk = 0;
            for (int q = 0; q < np_all; ++q) { p.piece_c16[(size_t)2 * q] = (int)nchunk; nchunk += ceil_div(p.piece_ptr[(size_t)q + 1] - p.piece_ptr[(size_t)q], CHL); }
            if (pack) {
                p.long_base.assign((size_t)nchunk, 0);
                p.long_cid16.resize((size_t)total);
                parallel_for(np_all, threads, 16, [&](long long q0, long long q1) {
                    for (long long q = q0; q < q1; ++q) {
                        const int e0 = p.piece_ptr[(size_t)q], e1 = p.piece_ptr[(size_t)q + 1], c0 = p.piece_c16[(size_t)2 * q];
                        bool narrow = true;
                        for (int e = e0, c = c0; e < e1; e += CHL, ++c) {
                            int lo = 2147483647, hi = -1;
                            for (int j = e; j < std::min(e + CHL, e1); ++j) { const int col = p.long_cid[(size_t)j]; if (col >= 0) { lo = std::min(lo, col); hi = std::max(hi, col); } }
                            p.long_base[(size_t)c] = hi < 0 ? 0 : lo;
                            if (hi >= 0 && (long long)hi - lo > 65534) narrow = false;
                        }
                        narrow = narrow && e1 - e0 >= kLong16MinChunks * CHL;
                        p.piece_c16[(size_t)2 * q + 1] = narrow ? 1 : 0;
                        for (int e = e0, c = c0; e < e1; e += CHL, ++c)
                            for (int j = e; j < std::min(e + CHL, e1); ++j) {
                                const int col = p.long_cid[(size_t)j];
                                p.long_cid16[(size_t)j] = !narrow ? (uint16_t)0 : col < 0 ? kLongPad16 : (uint16_t)(col - p.long_base[(size_t)c]);
                            }
                    }
                });
            } else { p.long_base.clear(); p.long_cid16.clear(); }
            p.cnt_long_chunks = (size_t)nchunk;
        }
    }

    lap("long rows");
    // ---- medium rows: regular tiles kept while a 16 x K chunk is >= threshold full
    // (the reference's rule, dasp_f64.h:1044-1091, on this geometry's tile), rest = irregular tail
    const int K = geo.med_k, CH = geo.chunk, VPL = geo.chunk / 64;      // values of one chunk per lane: 1 (f64) / 4 (f16)
    const int nb = ceil_div(nmed, kMedRows);
    std::vector<int> nchunks((size_t)nb + 1, 0);
    p.irr_ptr.assign((size_t)nmed + 1, 0);
    // 16-bit column ids: a chunk's ids are stored as u16 offsets from the chunk's smallest column (one int32 base per
    // chunk), 10 instead of 12 bytes per f64 nonzero (4 instead of 6 for f16).  A chunk whose columns span more than
    // 65534 cannot be stored that way; in cid16 mode the regular part of its block ends there and the rest of those rows
    // joins the irregular tail (32-bit ids).  opt.cid16: 0 = auto (on when that costs < 3 % of the regular elements).
    std::vector<int> nchunks16((size_t)nb + 1, 0);
    const bool try16 = p.opt.cid16 >= 0 && !meta_only;
    // one-byte ids (plan.hpp med_cid8): bit c of narrow_mask[b] = chunk c (< 64) of block b spans <= 254 columns.  f64 only: the f16 kernels
    // are not bound by bytes (profiles/r02_cid8.md)
    const bool try8 = try16 && !f16 && p.opt.cid8 >= 0;
    std::vector<unsigned long long> narrow_mask(try8 ? (size_t)nb : 0, 0ull);
    parallel_for(nb, threads, 256, [&](long long b0, long long b1) {
        for (long long b = b0; b < b1; ++b) {
            const int r0 = (int)b * kMedRows, r1 = std::min(nmed, r0 + kMedRows);
            int k = 0;
            for (;; ++k) {
                int fill = 0;
                for (int r = r0; r < r1; ++r) fill += std::min(K, std::max(0, lenM[r] - K * k));
                if (!(fill >= threshold * CH) || fill == 0) break;
            }
            nchunks[b] = k;
            int k16 = k;
            unsigned long long mask = 0;
            if (try16 && !dev) {
                for (int c = 0; c < k; ++c) {
                    int lo = 2147483647, hi = -1;
                    for (int r = r0; r < r1; ++r) {
                        const int a0 = rp[ridM[r]], i0 = c * K, i1 = std::min(lenM[r], i0 + K);
                        for (int i = i0; i < i1; ++i) { const int col = remap(ci[a0 + i]); lo = std::min(lo, col); hi = std::max(hi, col); }
                    }
                    if (hi >= 0 && (long long)hi - lo > 65534) { k16 = c; break; }
                    if (c < 64 && hi >= 0 && hi - lo <= 254) mask |= 1ull << c;
                }
            }
            nchunks16[b] = k16;
            if (try8 && !dev) narrow_mask[(size_t)b] = mask;
        }
    });
    if (try16 && dev) { if (int rc = devpack_chunk_spans(p, *dev, ridM, lenM, nchunks, nchunks16.data(), try8 ? narrow_mask.data() : nullptr)) return rc; }
    {
        long long e32 = 0, e16 = 0;
        for (int b = 0; b < nb; ++b) { e32 += nchunks[b]; e16 += nchunks16[b]; }
        // auto: not for small or LDS-windowed matrices (latency-bound; there the extra per-chunk base load costs more than the
        // 2 bytes per nonzero save: cop20k_A 11.6 vs 12.4 us).  From ~64 MiB of CSR on it pays on every FEM-like stand-in
        // (nlpkkt160 x0.03 16.7 -> 15.5 us, Queen x0.05 31.2 -> 27.7, x0.08 56.8 -> 42.5: the packed matrix then fits the
        // Infinity Cache); windowed plans keep the old bound.
        // r6 (tests/test_zz_auto_rules.py): from 16 MiB on -- nlpkkt160 x0.01 f64 (26 MB) 5.9 -> 5.4 us, Queen_4147 x0.03 f16 (56 MB) 12.1 -> 10.2 with them
        const bool streams = (long long)nnz * (geo.vbytes + 4) > (p.windowed ? kStreamBytes : (16ll << 20));
        // r3: plans of at most 256 LDS windows (one window workgroup per CU: the 128-register kernel) whose ids can be window-relative
        // (no per-chunk base load at all): cop20k_A 10.7 -> 10.4 us, 10.2 with the windows dealt to the XCDs in contiguous eighths
        // (r6: any number of windows -- the 64-register build no longer spills with 16-bit ids; cop20k_A x8 f64, 848 windows, 58.6 -> 45.6 us, x4 f16 18.4 -> 16.8)
        const bool rel1 = p.windowed && !p.win_hybrid && p.opt.x_window != -2 && p.lds_bytes / geo.vbytes <= 65534;
        p.cid16 = try16 && e32 > 0 && (p.opt.cid16 > 0 || ((streams || rel1) && (double)e16 >= 0.97 * (double)e32));
        if (p.cid16) nchunks.swap(nchunks16);
        // LDS-staged windows: offsets from the window's first staged column (every staged span is far below 65535 elements in f64;
        // f16 spans beyond that -- the 160 KiB cap -- keep the per-chunk bases)
        p.win_rel16 = p.cid16 && p.windowed && !p.win_hybrid && p.lds_bytes / geo.vbytes <= 65534;
    }
    parallel_for(nb, threads, 1 << 12, [&](long long b0, long long b1) {
        for (long long b = b0; b < b1; ++b) {
            const int r0 = (int)b * kMedRows, r1 = std::min(nmed, r0 + kMedRows);
            for (int r = r0; r < r1; ++r) p.irr_ptr[(size_t)r] = std::max(0, lenM[(size_t)r] - K * nchunks[(size_t)b]);
        }
    });
    // which blocks store chunk pairs (plan.hpp med_npair): none in a windowed plan; pipelined blocks and one-shot f16 blocks otherwise;
    // one-shot f64 blocks only when the plan is far beyond the 256 MiB Infinity Cache (nlpkkt160: 2.8 GB -7 % / -1.5 % by box; at 278 MB +4-8 %)
    p.pair_mode = p.windowed || p.opt.chunk_pairs < 0 ? 0 : p.opt.chunk_pairs > 0 ? std::min(2, p.opt.chunk_pairs)
                                                                     : ((long long)nnz * (geo.vbytes + 4) > 4 * kStreamBytes ? 2 : 1);
    // narrow chunks per block: those of its pipelined paired region, in whole pipeline batches.  Automatic (opt.cid8 = 0, late r5): not in ONE-SHOT blocks -- their one-byte ids are
    // 16-bit loads per lane and pair, slower per byte than the 16-bit ids they replace (27-point stencil on 160^3: 0.719 -> 0.830 of the roofline without them; nlpkkt160 with 98 % of its
    // chunks narrow 15-19 % slower, profiles/r05_id_encoding.md 4) -- and not at all when under a tenth of the chunks qualify: a plan with ANY narrow chunk runs the one-byte-id
    // kernel instantiation (80 registers; rows of 40 with 240 narrow chunks of 1.3 M: 0.740 against 0.764).  opt.cid8 = 1 keeps both (the r4 behaviour)
    std::vector<int> n8of((size_t)nb + 1, 0);
    if (try8 && p.cid16 && p.pair_mode > 0) {
        long long total8 = 0, total = 0;
        for (int b = 0; b < nb; ++b) {
            const int nt = (p.irr_ptr[(size_t)b * kMedRows] + K - 1) / K, npair = med_npair(nchunks[b], nt, geo.vbytes, p.pair_mode);
            const unsigned long long in = npair >= 64 ? ~0ull : ((1ull << npair) - 1);
            n8of[b] = p.opt.cid8 == 0 && med_oneshot64(nchunks[b], nt) ? 0 : med_n8(__builtin_popcountll(narrow_mask[(size_t)b] & in), nchunks[b], nt);
            total8 += n8of[b]; total += nchunks[b];
        }
        if (p.opt.cid8 == 0 && total8 * 10 < total) std::fill(n8of.begin(), n8of.end(), 0);
        // ... and (r6) only from 64 MiB of CSR on, where the 16-bit ids used to start and the one-byte ids were tuned: below it the launch is latency-bound and the 80-register
        // build with its longer id decode loses (HV15R x0.01, 30 MB: 8.1 -> 9.1 us when the lower 16-bit bound of r6 brought them along: tests/test_zz_auto_rules.py)
        if (p.opt.cid8 == 0 && p.opt.cid16 == 0 && (long long)nnz * (geo.vbytes + 4) <= (64ll << 20)) std::fill(n8of.begin(), n8of.end(), 0);      // (a caller who forces the 16-bit ids on a small matrix keeps the r5 rule)
    }
    lap("chunk split (+cid16 spans)");
    p.med_ptr.assign((size_t)nb + 1, 0);
    {
        long long run = 0;
        for (int b = 0; b < nb; ++b) { p.med_ptr[b] = (int)run; run += nchunks[b]; }
        if (run >= (1LL << 31) / 1) { set_error("too many medium chunks"); return DASP_ERR_ARG; }
        p.med_ptr[nb] = (int)run;
        // exclusive scan of the tail lengths: per-range sums, their prefix, then every range scans itself
        const int sparts = (int)std::max<long long>(1, std::min<long long>(threads, ((long long)nmed + (1 << 16) - 1) >> 16));
        std::vector<long long> psum((size_t)sparts + 1, 0);
        parallel_for(sparts, sparts, 1, [&](long long p0, long long p1) {
            for (long long q = p0; q < p1; ++q) {
                long long a = 0;
                for (long long r = (long long)nmed * q / sparts, e = (long long)nmed * (q + 1) / sparts; r < e; ++r) a += p.irr_ptr[(size_t)r];
                psum[(size_t)q + 1] = a;
            }
        });
        for (int q = 0; q < sparts; ++q) psum[(size_t)q + 1] += psum[(size_t)q];
        parallel_for(sparts, sparts, 1, [&](long long p0, long long p1) {
            for (long long q = p0; q < p1; ++q) {
                long long t = psum[(size_t)q];
                for (long long r = (long long)nmed * q / sparts, e = (long long)nmed * (q + 1) / sparts; r < e; ++r) { const int v = p.irr_ptr[(size_t)r]; p.irr_ptr[(size_t)r] = (int)t; t += v; }
            }
        });
        long long t = psum[(size_t)sparts];
        p.irr_ptr[nmed] = (int)t;
    }
    const long long n_reg = (long long)p.med_ptr[nb] * CH;
    const int nnz_irreg = p.irr_ptr[nmed];
    p.cnt_reg = (size_t)n_reg; p.cnt_irr = (size_t)nnz_irreg;
    p.med_base.assign(p.cid16 && pack ? (size_t)p.med_ptr[nb] : 0, 0);
    p.med_c8ptr.clear(); p.med_korig.clear(); p.med_cid8.clear(); p.cnt_reg8 = 0;
    if (p.cid16) {
        p.med_c8ptr.assign((size_t)nb + 1, 0);
        for (int b = 0; b < nb; ++b) p.med_c8ptr[(size_t)b + 1] = p.med_c8ptr[b] + n8of[b];
        p.cnt_reg8 = (size_t)p.med_c8ptr[nb] * CH;
        p.med_korig.assign((size_t)p.med_ptr[nb], 0);
    }
    if (pack) {
    p.med_val.resize((size_t)n_reg * sizeof(T));               // not zero-filled: each block pads its own region first
    p.med_cid.resize(p.cid16 ? 0 : (size_t)n_reg);
    p.med_cid16.resize(p.cid16 ? (size_t)n_reg - p.cnt_reg8 : 0);
    p.med_cid8.resize(p.cnt_reg8);
    p.irr_val.resize((size_t)nnz_irreg * sizeof(T));           // fully covered by the rows' tails
    p.irr_cid.resize((size_t)nnz_irreg);
    }
    if (pack) {
        T *mv = reinterpret_cast<T *>(p.med_val.data());
        T *iv = reinterpret_cast<T *>(p.irr_val.data());
        parallel_for(nb, threads, 64, [&](long long b0, long long b1) {
            std::vector<int> pos_of, lo_of;
            for (long long b = b0; b < b1; ++b) {
                const int nc = p.med_ptr[b + 1] - p.med_ptr[b];
                const size_t base = (size_t)p.med_ptr[b] * CH;
                const int r0 = (int)b * kMedRows, r1 = std::min(nmed, r0 + kMedRows);
                // tail steps of the block = those of its first (longest) row; with them the kernel decides one shot / pipeline, hence the layout
                const int nt_b = (p.irr_ptr[r0 + 1] - p.irr_ptr[r0] + K - 1) / K;
                const int npair = med_npair(nc, nt_b, geo.vbytes, p.pair_mode);
                const bool oneshot = med_oneshot64(nc, nt_b);          // (f64: the layout of the block's one-byte ids)
                // cid16 mode: position q of the block holds the block's chunk korig[q]: n8 narrow chunks (columns span <= 254: one-byte ids) of
                // the paired region first, everything else behind them in its original order
                const int n8 = p.cid16 ? p.med_c8ptr[b + 1] - p.med_c8ptr[b] : 0;
                const size_t e8 = p.cid16 ? (size_t)p.med_c8ptr[b] * CH : 0;                               // the block's narrow ids in med_cid8
                const size_t e16 = p.cid16 ? ((size_t)p.med_ptr[b] - (size_t)p.med_c8ptr[b]) * CH : 0;      // its wide ids in med_cid16
                pos_of.assign((size_t)nc + 1, 0); lo_of.assign((size_t)nc + 1, 0);
                if (p.cid16) {
                    int a8 = 0, a16 = 0;
                    for (int c = 0; c < nc; ++c) {
                        int lo = 2147483647, hi = -1;
                        for (int r = r0; r < r1; ++r) {
                            const int a0 = rp[ridM[r]], i0 = c * K, i1 = std::min(lenM[r], i0 + K);
                            for (int i = i0; i < i1; ++i) { const int col = remap(ci[a0 + i]); lo = std::min(lo, col); hi = std::max(hi, col); }
                        }
                        const bool narrow = c < npair && c < 64 && a8 < n8 && hi >= 0 && hi - lo <= 254;
                        lo_of[c] = lo == 2147483647 ? 0 : lo;
                        if (p.win_rel16) { const size_t w = (size_t)b / (size_t)(p.row_window / kMedRows); if (p.win_len[w] > 0) lo_of[c] = p.win_cmin[w]; }
                        pos_of[c] = c >= npair ? c : narrow ? a8++ : n8 + a16++;
                    }
                    for (int c = 0; c < nc; ++c) {
                        p.med_base[(size_t)p.med_ptr[b] + pos_of[c]] = lo_of[c];
                        p.med_korig[(size_t)p.med_ptr[b] + pos_of[c]] = c;
                    }
                } else for (int c = 0; c < nc; ++c) pos_of[c] = c;
                {   // pad the block's region (value 0, id -1 / 0xFFFF / 0xFF); real entries overwrite below
                    const size_t n = (size_t)nc * CH;
                    std::fill(mv + base, mv + base + n, (T)0);
                    if (p.cid16) {
                        std::fill(p.med_cid8.begin() + e8, p.med_cid8.begin() + e8 + (size_t)n8 * CH, (uint8_t)0xFF);
                        std::fill(p.med_cid16.begin() + e16, p.med_cid16.begin() + e16 + (size_t)(nc - n8) * CH, (uint16_t)0xFFFF);
                    } else std::fill(p.med_cid.begin() + base, p.med_cid.begin() + base + n, -1);
                }
                for (int r = r0; r < r1; ++r) {
                    const int rr = r - r0, row = ridM[r], len = lenM[r], a0 = rp[row];
                    const int nreg = std::min(len, nc * K);
                    for (int i = 0; i < nreg; ++i) {
                        const int c = i / K, kk = i % K, q = pos_of[c];
                        // f64: lane = kk*16 + rr holds A[rr][kk]            (one value per lane)
                        // f16: lane = (kk/4)*16 + rr holds A[rr][4*(kk/4)..+3] (four values per lane)
                        // a pipelined block's leading chunks are stored in pairs, [pair][lane][2 chunks][VPL] (plan.hpp med_npair)
                        const int lane = f16 ? (kk / 4) * kMedRows + rr : kk * kMedRows + rr, j = f16 ? kk % 4 : 0;
                        const size_t at = base + med_elem_index(npair, q, lane, j, VPL, CH);
                        mv[at] = val[a0 + i];
                        const int col = remap(ci[a0 + i]);
                        if (!p.cid16) p.med_cid[at] = col;
                        else if (q < n8) p.med_cid8[e8 + med_cid8_index(q, lane, CH, oneshot)] = (uint8_t)(col - lo_of[c]);
                        else p.med_cid16[e16 + med_elem_index(npair - n8, q - n8, lane, j, VPL, CH)] = (uint16_t)(col - lo_of[c]);
                    }
                    const int t0 = p.irr_ptr[r], tl = p.irr_ptr[r + 1] - t0;
                    for (int j = 0; j < tl; ++j) {   // the LAST tl entries of the row (dasp_f64.h:1094-1106)
                        iv[t0 + j] = val[a0 + len - tl + j];
                        p.irr_cid[t0 + j] = remap(ci[a0 + len - tl + j]);
                    }
                }
            }
        });
    }

    lap("pack medium");
    // ---- short rows: one slab per length, tile-major [tile][k][short_rows] -- or, for the rows of 1..4 nonzeros, the wave-segmented layout
    // (opt.short_seg, plan.hpp short_elem_index): auto = f64 plans without x windows (the windowed kernels are held to 64 registers and
    // spill more with the DPP path compiled in: cop20k_A 11.1 -> 12.0 us; f16 gains nothing: DESIGN.md section 3)
    {
        const int SR = geo.short_rows;
        const bool seg = p.opt.short_seg > 0 || (p.opt.short_seg == 0 && !f16 && !p.windowed);
        int seg_max_len = 4;          // the longest rows that take the segmented layout
        if (seg && p.opt.short_seg == 0 && !meta_only && nnz_short <= (16ll << 20)) {
            // r6 (tests/test_zz_auto_rules.py, tools/category_sweep.py): in a slab a lane owns whole rows, so the 64 lanes of a step gather for 128 neighbouring rows of one
            // length -- where neighbouring short rows read neighbouring columns (road networks, meshes' boundary rows) those gathers coalesce, and while the launch is
            // latency-bound the slabs win: rows of 1..4 within +-64 columns of the diagonal, 262 k / 1 M / 4 M rows: 4.9 / 15.5 / 50.2 us segmented against 3.9 / 12.3 / 46.9
            // as slabs.  At HBM scale the segmented tiles (256 elements per wave against 128 L) are the faster stream again (16 M rows: 219 against 192 us; 24 M: 356 / 325
            // with slabs for lengths 3 and 4 only) -- hence the bound of 16 M short nonzeros.  Graph rows (webbase-1M f64: 30.6 -> 28.4 us segmented) keep the segmented
            // tiles at every size.  The measure of the slab rule above -- pairs of successive rows of one length -- with a wider test: positions whose columns differ by < 256
            // (a tile's 128 rows then read a few KB of x: L1 hits, whatever the order inside)
            std::vector<int> pairs;
            for (int g = 0; g < 4; ++g) {
                const std::vector<int> &list = *glist[g];
                const int step = std::max(2, ((int)list.size() / 2048) & ~1);
                for (size_t i = 0; i + 1 < list.size(); i += (size_t)step) { pairs.push_back(list[i]); pairs.push_back(list[i + 1]); }
            }
            long long near = 0, entries = 0;
            if (dev) { if (int rc = devpack_row_coherence(p, *dev, pairs, 256, &near, &entries)) return rc; }
            else
                for (size_t q = 0; q + 1 < pairs.size(); q += 2) {
                    const int a = rp[pairs[q]], b = rp[pairs[q + 1]], len = rp[pairs[q] + 1] - a;
                    for (int k = 0; k < len; ++k) { const int d = remap(ci[a + k]) - remap(ci[b + k]); near += d > -256 && d < 256; }
                    entries += len;
                }
            if (entries >= 256 && 2 * near >= entries) seg_max_len = 0;
        }
        long long off = 0; int tile0 = 0;
        for (int g = 0; g < kNumShortGroups; ++g) {
            ShortGroup &G = p.grp[g];
            G.seg = seg && !p.windowed && G.len >= 1 && G.len <= seg_max_len && g < 5 ? 1 : 0;
            G.rpt = G.seg ? short_seg_rows(G.len) : SR;
            G.tiles = ceil_div(G.count, G.rpt);
            G.tile0 = tile0; G.elem_off = off;
            tile0 += G.tiles;
            off += (long long)G.tiles * short_tile_elems(G.seg != 0, G.len, SR);
        }
        if (off >= (1LL << 40)) { set_error("short segment too large"); return DASP_ERR_ARG; }
        p.cnt_short = (size_t)off;
        if (pack) {
        p.short_val.resize((size_t)off * sizeof(T));           // not zero-filled: only a slab's last tile has pads
        p.short_cid.resize((size_t)off);
        T *sv = reinterpret_cast<T *>(p.short_val.data());
        for (int g = 0; g < kNumShortGroups; ++g) {
            const ShortGroup &G = p.grp[g];
            if (G.tiles == 0 || G.len == 0) continue;
            // pads: a slab's last tile only; a segmented group of 3 has an idle lane in every 16, so all of it is cleared first
            const size_t te = (size_t)short_tile_elems(G.seg != 0, G.len, SR);
            const size_t t0 = (size_t)G.elem_off + (G.seg && G.len == 3 ? 0 : (size_t)(G.tiles - 1) * te), t1 = (size_t)G.elem_off + (size_t)G.tiles * te;
            std::fill(sv + t0, sv + t1, (T)0);
            std::fill(p.short_cid.begin() + t0, p.short_cid.begin() + t1, -1);
        }
        for (int g = 0; g < kNumShortGroups; ++g) {
            const ShortGroup &G = p.grp[g];
            if (G.len == 0 || G.count == 0) continue;
            const std::vector<int> &list = *glist[g];
            parallel_for(G.count, threads, 1 << 14, [&](long long b, long long e) {
                for (long long t = b; t < e; ++t) {
                    const int row = list[t], a0 = rp[row];
                    for (int k = 0; k < G.len; ++k) {
                        const size_t at = (size_t)G.elem_off + short_elem_index(G.seg != 0, G.len, SR, t, k);
                        sv[at] = val[a0 + k];
                        p.short_cid[at] = remap(ci[a0 + k]);
                    }
                }
            });
        }
        }   // pack
    }

    lap("pack short");
    // ---- stats: the reference's CSV counters (dasp_f64.h:1439-1441) + native sizes
    dasp_stats_t &s = p.stats;
    std::memset(&s, 0, sizeof s);
    s.precision = p.precision; s.rowA = m; s.colA = p.n; s.nnzA = nnz;
    s.short_row_1 = n1; s.common_13 = c13; s.short_row_3 = n3; s.short_row_4 = n4; s.short_row_2 = n2;
    s.short_seg = 0;
    for (int g = 0; g < kNumShortGroups; ++g) s.short_seg |= p.grp[g].seg;
    s.row_long = nlong_cls; s.row_block = nmed_all; s.row_zero = nz0; s.med_rows_as_pieces = nsp; s.chunk_pairs = p.pair_mode; s.cid8_chunks = p.med_c8ptr.empty() ? 0 : p.med_c8ptr.back();
    s.nnz_short = nnz_short; s.nnz_long = (int)nnz_long; s.nnz_irreg = nnz_irreg;
    s.origin_nnz_reg = nnz - nnz_irreg - (int)nnz_long - nnz_short;
    s.rowloop = nmed < 59990 ? 1 : (nmed < 400000 ? 2 : 4);
    s.fill0_nnz_short = (long long)p.cnt_short;
    s.fill0_nnz_long = (long long)p.cnt_long;
    s.fill0_nnz_reg = n_reg;
    const long long stored = s.fill0_nnz_short + s.fill0_nnz_long + s.fill0_nnz_reg + nnz_irreg;
    s.rate_fill0 = nnz > 0 ? (double)(stored - nnz) / nnz : 0.0;
    const long long sv = geo.vbytes;
    s.cid16_on = p.cid16 ? 1 : 0;
    s.data_X = (long long)(m + p.n) * sv + stored * (sv + 4) - (p.cid16 ? 2 * n_reg - 4ll * p.med_ptr[nb] : 0) - (long long)p.cnt_reg8 +
               (long long)(p.piece_ptr.size() + p.piece_dst.size() + p.multi_ptr.size() + p.multi_dst.size()) * 4 +
               (long long)(p.med_ptr.size() + p.irr_ptr.size()) * 4 + (natural ? (long long)m * 4 : 0) +
               (long long)(p.med_dst.size() + 2 * p.win_len.size()) * 4;
    s.data_origin1 = (long long)(nnz + p.n + m) * sv + (long long)nnz * 4 + (long long)(m + 1) * 4;  // main_f64.cu:143
    s.ref_fill0_nnz_short = refg.fill0_short; s.ref_fill0_nnz_long = refg.fill0_long; s.ref_fill0_nnz_reg = refg.fill0_reg; s.ref_data_X = refg.data_X;
    s.ref_nnz_irreg = refg.nnz_irreg; s.ref_origin_nnz_reg = nnz - refg.nnz_irreg - (int)nnz_long - nnz_short; s.ref_blocknum = refg.blocknum;
    s.ref_warp_number = refg.warp_number; s.ref_rate_fill0 = refg.rate_fill0;
    s.n_med_blocks = nb;
    s.n_long_pieces = (int)p.piece_dst.size();
    s.n_long_multi = (int)p.multi_dst.size();
    s.n_short_tiles = 0;
    for (int g = 0; g < kNumShortGroups; ++g) s.n_short_tiles += p.grp[g].tiles;
    s.n_workgroups = ceil_div(s.n_long_pieces, kWavesPerWG) + ceil_div(nb, kWavesPerWG) + ceil_div(s.n_short_tiles, kWavesPerWG);
    s.x_window_on = p.windowed ? 1 : 0; s.n_windows = (int)p.win_len.size(); s.row_window = p.row_window; s.lds_bytes = p.lds_bytes;
    s.n_windows_lds = 0;
    for (int v : p.win_len) s.n_windows_lds += v > 0;
    s.window_nnz_frac = window_frac;
    s.x_window_hybrid = p.win_hybrid ? 1 : 0;
    if (p.windowed) { const int wpw = std::min(16, p.row_window / kMedRows); s.n_workgroups = ceil_div(s.n_long_pieces, wpw) + ceil_div(s.n_windows, 8) * 8 + (win_fold_tiles(s.n_windows, s.n_short_tiles) ? 0 : ceil_div(s.n_short_tiles, wpw)); }
    if (dev && !meta_only) {   // the O(nnz) copies happen on the GPU; the plan comes back uploaded
        PackMeta meta;
        meta.ridL = &ridL; meta.startL = &startL; meta.ridM = &ridM; meta.lenM = &lenM;
        for (int g = 0; g < kNumShortGroups; ++g) meta.glist[g] = glist[g];
        if (int rc = devpack_all(p, *dev, meta)) return rc;
        lap("device pack");
    }
    if (meta_only) {   // the panels own the packed data; this plan keeps order_rid + the whole-matrix counters
        p.cnt_long = p.cnt_reg = p.cnt_irr = p.cnt_short = 0;
        p.piece_ptr.clear(); p.piece_dst.clear(); p.multi_ptr.clear(); p.multi_dst.clear(); p.piece_c16.clear(); p.long_base.clear(); p.long_cid16.clear(); p.cnt_long_chunks = 0;
        p.med_ptr.clear(); p.irr_ptr.clear();
        for (int g = 0; g < kNumShortGroups; ++g) { p.grp[g].tiles = 0; p.grp[g].tile0 = 0; p.grp[g].elem_off = 0; }
    }
    s.pre_ms = std::chrono::duration<double, std::milli>(clk::now() - t_begin).count();
    return DASP_OK;
}

// ---- column panels (opt.col_panels; DESIGN.md section 4 "column panels") -------------------------------------------------
// auto rule: only matrices whose rows scatter over more x than an XCD's L2 holds gain from cache blocking; anything with
// locality (FEM / stencil rows touch runs of neighbouring columns, banded rows stay inside a narrow span) is left alone.
// *scattered (r5; what the two-phase rule asks): 1 when a matrix of >= 10 M nonzeros has rows whose gathers scatter -- > 50 % of a sampled row's nonzeros on distinct
// 128-byte lines of x, a third of the entries in rows spanning > x / 4 -- whatever the size of x and however hot some of its lines are
static int decide_panels(const Plan &p, const int *rp, const int *ci, const Remap &remap, const DevCsr *dev, int *scattered)
{
    if (scattered) *scattered = 0;
    const int want = p.opt.col_panels;
    if (want > 64) { set_error("col_panels must be <= 64"); return DASP_ERR_ARG; }
    if (!p.dst_map.empty()) return 1;
    if (want == 1 || want < 0) return 1;
    if (want >= 2) return p.nnz > 0 && p.m > 0 ? want : 1;
    const long long xlen = p.opt.n_parts > 0 ? (long long)p.opt.n_parts * p.opt.part_stride : (long long)p.n;
    const long long vb = p.geo.vbytes, xbytes = xlen * vb;
    // (the two-phase form pays from ~10 M nonzeros on: rmat_2M x0.25 / x0.3 / x0.5 = 8.4 / 9.0 / 16.8 M: 0.0363 / 0.0354 / 0.0556 ms against 0.0323 / 0.0377 / 0.0609;
    // webbase-1M x4 = 14.3 M: 0.0527 against 0.0651; cache blocking keeps its 16 M)
    if (p.nnz < (10 << 20) || p.m <= 0) return 1;
    // the samples below cost a pass over 4096 rows (a gather kernel + a copy for a device CSR): skipped where no outcome could use them --
    // the two-phase rule cannot fire (f64, a column remap, a panel, or the caller chose), and either the matrix is below the panels' 16 M nonzeros or its x fits an
    // XCD's L2 and the O(rows) hub-row test (the only way to panels then) already says no
    const bool tp_possible = scattered && p.precision == 16 && p.opt.n_parts == 0 && p.opt.two_phase == 0 && !p.panel;
    auto hub_panels = [&]() -> int {
        if (p.opt.long_cb < 0 || p.opt.n_parts > 0) return 1;
        const long long h = std::max<long long>(p.opt.block_longest, 64ll * ((p.n + (vb == 8 ? 16384 : 32768) - 1) / (vb == 8 ? 16384 : 32768)));
        long long hub = 0;
        for (int i = 0; i < p.m; ++i) { const int len = rp[i + 1] - rp[i]; if (len >= h) hub += len; }
        return hub * 2 >= (long long)p.nnz && hub < (long long)p.nnz ? 2 : 1;
    };
    if (!tp_possible) {
        if (p.nnz < (16 << 20)) return 1;
        if (xbytes <= (4ll << 20) && hub_panels() == 1) return 1;
    }
    const int line_shift = vb == 8 ? 4 : 6;                       // 128-byte lines of x
    const int S = 4096;
    long long entries = 0, lines = 0, wide = 0;
    std::vector<long long> core_spans;          // per sampled row: 10th to 90th percentile of its columns
    std::vector<int> cols;
    // device CSR (r3): the sampled rows' column ids are gathered by a kernel and copied over (<= 2 M ids), the rule itself is the same
    std::vector<int> fetched;
    if (dev) {
        std::vector<long long> idx;
        for (int s = 0; s < S; ++s) {
            const int r = (int)((long long)p.m * s / S), len = rp[r + 1] - rp[r];
            if (len < 4) continue;
            for (int j = 0; j < std::min(len, 512); ++j) idx.push_back((long long)rp[r] + j);
        }
        if (int rc = devpack_gather_columns(p, *dev, &idx, 0, 0, 0, fetched)) return rc;
    }
    size_t fpos = 0;
    for (int s = 0; s < S; ++s) {
        const int r = (int)((long long)p.m * s / S);
        const int len = rp[r + 1] - rp[r];
        if (len < 4) continue;
        const int take = std::min(len, 512);
        cols.resize((size_t)take);
        if (dev) { for (int j = 0; j < take; ++j) cols[j] = fetched[fpos + (size_t)j]; fpos += (size_t)take; }
        else for (int j = 0; j < take; ++j) cols[j] = remap(ci[rp[r] + j]);
        std::sort(cols.begin(), cols.end());
        int distinct = 1;
        for (int j = 1; j < take; ++j) distinct += (cols[j] >> line_shift) != (cols[j - 1] >> line_shift);
        entries += take; lines += distinct;
        if ((long long)(cols[take - 1] - cols[0]) * vb > xbytes / 4) wide += take;
        core_spans.push_back((long long)cols[take - 1 - take / 10] - cols[take / 10] + 1);
    }
    if (entries < 4096) return 1;
    // the two-phase form pays from a weaker scatter on (rmat_2M f16: 0.69 lines per entry, 49 % of the entries in wide rows: 0.135 -> 0.094 ms) than cache blocking does
    // r6 (tests/test_zz_auto_rules.py): not scattered where the L1 serves the gathers anyway -- rows of (nearly) one length, which keep their neighbours under the length
    // sort, whose CORE spans under 16 KB of x however far a few of their entries reach: 1 M rows of 12 within +-500 columns + 10 % anywhere in f16 took the two-phase form
    // and ran 35.2 us against 22.6 as plain blocks.  (webbase-like rows -- many lengths, so blocks of unrelated rows -- stay scattered: x4 51.1 against 61.8 us two-phase.)
    bool l1_served = false;
    if (!core_spans.empty()) {
        std::nth_element(core_spans.begin(), core_spans.begin() + (long long)(core_spans.size() / 2), core_spans.end());
        if (core_spans[core_spans.size() / 2] * vb < 16 * 1024) {
            std::vector<long long> by_len(257, 0);
            long long med_nnz = 0;
            for (int i = 0; i < p.m; ++i) { const int len = rp[i + 1] - rp[i]; if (len >= 5 && len < p.opt.block_longest && len <= 256) { by_len[(size_t)len] += len; med_nnz += len; } }
            l1_served = med_nnz > 0 && 5 * *std::max_element(by_len.begin(), by_len.end()) >= 4 * med_nnz && 2 * med_nnz >= (long long)p.nnz;
        }
    }
    if (scattered && !l1_served && (double)lines > 0.5 * (double)entries && (double)wide >= 0.33 * (double)entries) *scattered = 1;
    if (p.nnz < (16 << 20)) return 1;
    if ((double)lines <= 0.75 * (double)entries || (double)wide < 0.5 * (double)entries) return 1;
    if (xbytes <= (4ll << 20)) {      // x fits an XCD's L2: nothing to block (A/B: 2.9 MB loses 10 %, 4.4 MB wins 11 %) ...
        // ... but hub rows that hold most of the nonzeros gain from the column-blocked form, which lives in a panel plan (longcb.cpp: x staged in LDS instead of
        // one L1 miss per nonzero): two panels for its sake (r5: powerlaw_1M x0.3 f64, 27 M nonzeros, x = 2.5 MB: 0.141 -> 0.114 ms; 0.145 with the hubs left to the panels)
        return hub_panels();
    }
    {   // scattered is not enough: real graphs have popular columns, and if the hottest 3 MiB of x lines already take most of
        // the gathers the L2 serves them without blocking (R-MAT 2^21 f64: 92 % of the gathers on 3 MiB of lines, panels
        // -24 %; 2^23: 76 %, panels +9 %; the uniform stand-ins: 25-44 %, panels +45-55 %)
        const long long xlines = (xlen >> line_shift) + 1;
        std::vector<int> hist((size_t)xlines, 0);
        const long long S2 = std::min<long long>(p.nnz, 1ll << 21), stride = std::max<long long>(1, p.nnz / S2);
        long long taken = 0;
        if (dev) {
            const long long cnt = (p.nnz + stride - 1) / stride;
            if (int rc = devpack_gather_columns(p, *dev, nullptr, 0, stride, cnt, fetched)) return rc;
            for (int c : fetched) { hist[(size_t)(c >> line_shift)]++; ++taken; }
        } else
        for (long long j = 0; j < p.nnz; j += stride) { hist[(size_t)(remap(ci[j]) >> line_shift)]++; ++taken; }
        const size_t cap = (size_t)((3ll << 20) / 128);
        long long hot = 0;
        if (hist.size() > cap) {
            std::nth_element(hist.begin(), hist.begin() + (long long)cap, hist.end(), std::greater<int>());
            for (size_t k = 0; k < cap; ++k) hot += hist[k];
        } else hot = taken;
        if ((double)hot >= 0.8 * (double)taken) return 1;
    }
    // panels of ~2.75 MiB of x (they must fit the 4 MiB L2 of an XCD next to the streamed tiles), and >= ~4 nonzeros per row
    // and panel: every extra panel re-pays the per-row cost (row tables, a partial y, shorter rows with fewer gathers in
    // flight).  Sweep on the stand-ins (tools/panel_probe.py): powerlaw_1M f64 (x 8 MB) 0.99 / 0.71 / 0.68 / 0.70 / 0.82 ms at
    // 1 / 2 / 3 / 4 / 8 panels, ljournal-2008 f16 (x 10.7 MB) 0.95 / 0.65 / 0.61 / 0.61 / 0.71 ms.
    const long long target = (11ll << 20) / 4;
    const long long by_x = (xbytes + target - 1) / target;
    const long long by_len = std::max<long long>(2, ((long long)p.nnz / p.m + 2) / 4);
    return (int)std::max<long long>(2, std::min<long long>(std::min(by_x, by_len), 8));
}

template <class T>
static int build_panels(Plan &p, const int *rp, const int *ci, const T *val, int P, const DevCsr *dev)
{
    const DevCsr *const dev_in = dev;       // (a device CSR whose long rows go column-blocked is fetched and split on the host: see long_cb below)
    using clk = std::chrono::steady_clock;
    const auto t_begin = clk::now();
    const bool verbose = std::getenv("DASP_VERBOSE") != nullptr;
    auto tick = t_begin;
    auto lap = [&](const char *what) { if (verbose) { auto now = clk::now(); std::fprintf(stderr, "[dasp panels] %-26s %.3f s\n", what, std::chrono::duration<double>(now - tick).count()); tick = now; } };
    // whole-matrix classification: order_rid and the reference's counters are those of the unsplit matrix (row lengths only)
    if (int rc = build_impl<T>(p, rp, ci, val, nullptr, kMetaOnly)) return rc;
    lap("whole-matrix meta");
    const int m = p.m;
    const int threads = resolve_threads(p.opt.host_threads);
    Remap remap;
    remap.n_parts = p.opt.n_parts; remap.stride = p.opt.part_stride; remap.b = p.part_bounds.data();
    const long long xlen = p.opt.n_parts > 0 ? (long long)p.opt.n_parts * p.opt.part_stride : (long long)p.n;
    std::vector<int> bnd((size_t)P + 1);
    for (int k = 0; k <= P; ++k) bnd[k] = (int)std::min<long long>(xlen, ((xlen * k / P + 63) / 64) * 64);   // whole 128-byte lines
    bnd[0] = 0; bnd[P] = (int)xlen;
    auto panel_of = [&](int c) { return (int)(std::upper_bound(bnd.begin() + 1, bnd.end(), c) - bnd.begin()) - 1; };

    // split the CSR by column range: count, prefix, scatter (row-parallel; the rows' internal order is kept)
    std::vector<std::vector<int>> rpP((size_t)P);
    std::vector<raw_vector<int>> ciP((size_t)P);
    std::vector<raw_vector<T>> valP((size_t)P);
    std::vector<DevCsr> devP;                          // device CSR: the same split by two kernels (devpack.hip), sub-matrices stay on the GPU
    std::vector<std::shared_ptr<void>> dev_keep;
    std::vector<DevRowTiles> rt_dev((size_t)P);        // device path: the tiles' elements, moved into the panel plan's arena once it exists
    std::vector<std::vector<std::shared_ptr<void>>> rt_keep((size_t)P);
    const bool natural = p.opt.y_order == DASP_Y_NATURAL;
    std::vector<int> slot_of_row;
    if (!natural) { slot_of_row.resize((size_t)m); for (int i = 0; i < m; ++i) slot_of_row[p.order[i]] = i; }
    lap("slot table");
    // column-blocked long rows (opt.long_cb, plan.hpp struct LongCB): the hub rows leave the panels -- in_lcb[row] -- and get their own LDS-staged kernel
    std::vector<unsigned char> in_lcb;
    raw_vector<int> lcb_ci;
    raw_vector<T> lcb_val;
    p.lcb = LongCB{};
    if (decide_long_cb(p, rp, P, in_lcb) > 0) {
        if (dev) {      // their cut by column block runs on the host for now: fetch the nonzeros once, then the host split below
            lcb_ci.resize((size_t)p.nnz); lcb_val.resize((size_t)p.nnz);
            if (int rc = devpack_fetch_csr(p, *dev, lcb_ci.data(), lcb_val.data())) return rc;
            ci = lcb_ci.data(); val = lcb_val.data(); dev = nullptr;
        }
        const int rc_lcb = build_long_cb(p, rp, ci, val, in_lcb, natural ? nullptr : slot_of_row.data());
        if (rc_lcb == 1) in_lcb.clear();          // (a piece too long for the kernel's LDS slice: the rows stay in the panels)
        else if (rc_lcb) return rc_lcb;
        lap("long rows by column block");
    }
    const bool lcb_on = !in_lcb.empty();
    // row tiles (opt.row_tile_max, Plan::rt_*): panel k keeps its rows of <= rt_max nonzeros in the parent's output order; rt[k] holds their
    // tables, rt_at[k][row] = the row's first element in the tiles' arrays (-1: the row stays with the panel's own plan)
    int rt_max = p.opt.row_tile_max == 0 ? (p.precision == 16 ? kRowTileAuto : kRowTileAuto64) : std::max(0, p.opt.row_tile_max);
    if (rt_max > kRowTileMax) { set_error("row_tile_max must be <= 32"); return DASP_ERR_ARG; }
    if (p.opt.row_tile_max == 0 && ab_knobs().row_tile_max >= 0) rt_max = std::min(kRowTileMax, ab_knobs().row_tile_max);      // A/B knob: only over "auto", never over a caller's choice
    struct RowTiles { std::vector<int> ptr; std::vector<uint16_t> start; std::vector<uint64_t> mask; raw_vector<char> val; raw_vector<int> cid; std::vector<int> at; size_t cnt = 0; };
    std::vector<RowTiles> rt((size_t)P);
    // from a panel's row LENGTHS in len[1 .. m] (len[i + 1] = row i): the tiles' tables; the rows taken get length 0 in len
    auto cut_row_tiles = [&](RowTiles &R, int *len) {
        const int tiles = ceil_div(m, kRowTile);
        R.ptr.assign((size_t)tiles + 1, 0); R.start.assign((size_t)tiles * kRowTile, 0); R.mask.assign((size_t)tiles, 0); R.at.assign((size_t)m, -1);
        long long run = 0;
        for (int t = 0; t < tiles; ++t) {
            int in_tile = 0;
            for (int i = 0; i < kRowTile; ++i) {
                const int j = t * kRowTile + i;
                R.start[(size_t)j] = (uint16_t)in_tile;
                if (j >= m) continue;
                const int r = natural ? j : p.order[(size_t)j], L = len[(size_t)r + 1];
                if (L > rt_max) continue;
                R.mask[(size_t)t] |= uint64_t(1) << i;
                if (L > 0) { R.at[(size_t)r] = (int)(run + in_tile); in_tile += L; len[(size_t)r + 1] = 0; }
            }
            run += in_tile;
            R.ptr[(size_t)t + 1] = (int)run;
        }
        R.cnt = (size_t)run;
    };
    if (dev) {
        if (int rc = devpack_panel_split(p, *dev, bnd, P, rpP, devP, dev_keep)) return rc;
        if (rt_max > 0) {
            // the split's row pointers -> lengths, cut, -> the pointers of what is left; the device moves the elements (devpack.hip)
            parallel_for(P, threads, 1, [&](long long k0, long long k1) {
                for (long long k = k0; k < k1; ++k) {
                    int *q = rpP[k].data();
                    for (int i = m; i > 0; --i) q[i] -= q[i - 1];
                    cut_row_tiles(rt[(size_t)k], q);
                    for (int i = 0; i < m; ++i) q[i + 1] += q[i];
                }
            });
            // (the device moves each panel's elements in that panel's worker below, beside the other panels' builds)
        }
    } else {
    for (auto &v : rpP) v.assign((size_t)m + 1, 0);
    parallel_for(m, threads, 1 << 12, [&](long long b, long long e) {
        for (long long i = b; i < e; ++i) {
            if (lcb_on && in_lcb[(size_t)i]) continue;          // an empty row in every panel
            for (int j = rp[i]; j < rp[i + 1]; ++j) rpP[panel_of(remap(ci[j]))][i + 1]++;
        }
    });
    parallel_for(P, threads, 1, [&](long long k0, long long k1) {
        for (long long k = k0; k < k1; ++k) {
            int *q = rpP[k].data();
            if (rt_max > 0) cut_row_tiles(rt[(size_t)k], q);
            for (int i = 0; i < m; ++i) q[i + 1] += q[i];
        }
    });
    const size_t vb = (size_t)p.geo.vbytes;
    for (int k = 0; k < P; ++k) { ciP[k].resize((size_t)rpP[k][m]); valP[k].resize((size_t)rpP[k][m]); rt[(size_t)k].cid.resize(rt[(size_t)k].cnt); rt[(size_t)k].val.resize(rt[(size_t)k].cnt * vb); }
    parallel_for(m, threads, 1 << 12, [&](long long b, long long e) {
        std::vector<int> cur((size_t)P);
        std::vector<char> tiled((size_t)P);
        for (long long i = b; i < e; ++i) {
            if (lcb_on && in_lcb[(size_t)i]) continue;
            for (int k = 0; k < P; ++k) { const int at = rt_max > 0 ? rt[(size_t)k].at[(size_t)i] : -1; tiled[k] = at >= 0; cur[k] = at >= 0 ? at : rpP[k][i]; }
            for (int j = rp[i]; j < rp[i + 1]; ++j) {
                const int c = remap(ci[j]), k = panel_of(c), at = cur[k]++;
                if (tiled[k]) { rt[(size_t)k].cid[(size_t)at] = c; reinterpret_cast<T *>(rt[(size_t)k].val.data())[at] = val[j]; }
                else { ciP[k][at] = c; valP[k][at] = val[j]; }
            }
        }
    });
    }

    lap("split by column range");
    const bool streams = (long long)p.nnz * (p.geo.vbytes + 4) > kStreamBytes;
    p.panels.clear(); p.panel_bounds.clear();
    // the panels are built side by side (their O(rows) classifier passes are serial), each with its share of the threads
    std::vector<std::unique_ptr<dasp_plan>> built((size_t)P);
    std::vector<int> rcs((size_t)P, DASP_OK);
    std::vector<std::string> errs((size_t)P);
    {
        // device path too: a panel's O(rows) host stages overlap another panel's kernels and copies (ljournal-2008, 4 panels: 4 x 26 ms one after the other)
        const int side = dev ? std::min(P, std::min(4, threads)) : std::min(P, threads), each = std::max(1, threads / side);
        const int hip_device = dev ? devpack_current_device() : -1;
        std::atomic<int> next{0};
        std::vector<std::thread> workers;
        auto work = [&] {
                if (dev) devpack_use_device(hip_device);          // a new thread starts on device 0
                for (int k = next++; k < P; k = next++) {
                    const int nnz_k = rpP[k][m];
                    RowTiles &R = rt[(size_t)k];
                    if (nnz_k == 0 && R.cnt == 0) continue;         // an empty panel adds nothing
                    if (dev && rt_max > 0) {
                        rcs[k] = devpack_row_tiles(p, devP[(size_t)k], rpP[(size_t)k], R.at, R.cnt, rt_keep[(size_t)k], &rt_dev[(size_t)k]);
                        if (rcs[k] != DASP_OK) { errs[k] = last_error_cstr(); continue; }
                    }
                    std::unique_ptr<dasp_plan> h(new dasp_plan());
                    Plan &q = h->impl;
                    q.precision = p.precision; q.geo = p.geo; q.m = m; q.n = (int)xlen; q.nnz = nnz_k;
                    q.opt = p.opt;
                    q.opt.y_order = DASP_Y_NATURAL; q.opt.n_parts = 0; q.opt.part_bounds = nullptr; q.opt.part_stride = 0;
                    q.opt.col_panels = 1; q.opt.host_threads = each;
                    if (q.opt.stream_policy == 0) q.opt.stream_policy = streams ? 2 : 1;   // the policy follows the whole matrix, not one panel
                    q.dst_map = slot_of_row; q.panel = true;
                    if (R.cnt > 0) {      // (a panel none of whose rows is short enough keeps no tiles at all)
                        q.rt_max = rt_max; q.cnt_rt = R.cnt;
                        q.rt_ptr.swap(R.ptr); q.rt_start.swap(R.start); q.rt_mask.swap(R.mask); q.rt_val.swap(R.val); q.rt_cid.swap(R.cid);
                        q.opt.x_window = -1; q.opt.cid8 = -1;      // the tiles ride in the non-windowed kernel without one-byte ids (dasp_spmv_rt_kernel)
                    }
                    // f16 only: 2-byte stores are where the partial lines hurt (ljournal-2008-uniform 0.590 -> 0.559 ms, ljournal-2008 0.506 -> 0.503; powerlaw_1M f64 0.651 -> 0.654)
                    if (!natural && p.precision == 16 && !ab_knobs().panel_row_scan) q.scan_order = p.order.data();      // (DASP_PANEL_ROW_SCAN: A/B knob, the r3 order)
                    try { rcs[k] = build_impl<T>(q, rpP[k].data(), dev ? nullptr : ciP[k].data(), dev ? nullptr : valP[k].data(), dev ? &devP[(size_t)k] : nullptr, kPanel); }
                    catch (const std::bad_alloc &) { rcs[k] = DASP_ERR_NOMEM; set_error("out of host memory"); }
                    if (rcs[k] == DASP_OK && dev && q.cnt_rt > 0) rcs[k] = devpack_place_row_tiles(q, rt_dev[(size_t)k]);
                    if (rcs[k] != DASP_OK) { errs[k] = last_error_cstr(); continue; }
                    q.opt.host_threads = p.opt.host_threads; q.scan_order = nullptr;
                    {   // the tiles in the panel's counters
                        dasp_stats_t &t = q.stats;
                        t.row_tile_max = q.rt_max; t.n_row_tiles = (int)q.rt_mask.size(); t.row_tile_nnz = (long long)q.cnt_rt;
                        t.n_workgroups += ceil_div(t.n_row_tiles, kWavesPerWG);
                        t.data_X += (long long)q.cnt_rt * (p.geo.vbytes + 4) + (long long)t.n_row_tiles * (4 + 2 * kRowTile + 8);
                    }
                    std::vector<int>().swap(rpP[k]); raw_vector<int>().swap(ciP[k]); raw_vector<T>().swap(valP[k]);
                    built[k] = std::move(h);
                }
            };
        for (int t = 0; t < side; ++t) workers.emplace_back(work);
        for (auto &w : workers) w.join();
    }
    lap("panel plans");
    for (int k = 0; k < P; ++k) {
        if (rcs[k] != DASP_OK) { set_error(errs[k]); return rcs[k]; }
        if (!built[k]) continue;
        p.panels.push_back(std::move(built[k]));
        p.panel_bounds.push_back(bnd[k]); p.panel_bounds.push_back(bnd[k + 1]);
    }

    // native counters: sums over the panels (the classifier counters above stay those of the whole matrix)
    dasp_stats_t &s = p.stats;
    const long long vb = p.geo.vbytes;
    const int K = (int)p.panels.size();
    s.fill0_nnz_short = s.fill0_nnz_long = s.fill0_nnz_reg = 0;
    s.n_med_blocks = s.n_long_pieces = s.n_long_multi = s.n_short_tiles = s.n_workgroups = 0;
    s.x_window_on = s.n_windows = s.n_windows_lds = s.lds_bytes = s.row_window = s.cid16_on = s.x_window_hybrid = s.chunk_pairs = s.cid8_chunks = 0;
    s.row_tile_max = s.n_row_tiles = 0; s.row_tile_nnz = 0;
    s.window_nnz_frac = 0.0;
    long long stored = 0, dataX = 0;
    for (const auto &h : p.panels) {
        const dasp_stats_t &t = h->impl.stats;
        s.fill0_nnz_short += t.fill0_nnz_short; s.fill0_nnz_long += t.fill0_nnz_long; s.fill0_nnz_reg += t.fill0_nnz_reg;
        stored += t.fill0_nnz_short + t.fill0_nnz_long + t.fill0_nnz_reg + t.nnz_irreg;
        dataX += t.data_X - (long long)(m + h->impl.n) * vb;
        s.n_med_blocks += t.n_med_blocks; s.n_long_pieces += t.n_long_pieces; s.n_long_multi += t.n_long_multi;
        s.n_short_tiles += t.n_short_tiles; s.n_workgroups += t.n_workgroups;
        s.x_window_on |= t.x_window_on; s.n_windows += t.n_windows; s.n_windows_lds += t.n_windows_lds;
        s.lds_bytes = std::max(s.lds_bytes, t.lds_bytes); s.row_window = std::max(s.row_window, t.row_window);
        s.cid16_on |= t.cid16_on; s.x_window_hybrid |= t.x_window_hybrid; s.chunk_pairs = std::max(s.chunk_pairs, t.chunk_pairs); s.cid8_chunks += t.cid8_chunks;
        s.window_nnz_frac += t.window_nnz_frac * (double)t.nnzA / (double)std::max(1, p.nnz);
        s.row_tile_max = std::max(s.row_tile_max, t.row_tile_max); s.n_row_tiles += t.n_row_tiles; s.row_tile_nnz += t.row_tile_nnz; stored += t.row_tile_nnz;
    }
    s.rate_fill0 = p.nnz > 0 ? (double)(stored - p.nnz) / p.nnz : 0.0;
    // packed panels + x once + every panel's partial y written and read back + y
    s.data_X = dataX + xlen * vb + (long long)(2 * K + 1) * m * vb;
    s.n_col_panels = K;
    if (lcb_on) {
        const LongCB &L = p.lcb;
        s.lcb_rows = L.n_rows(); s.lcb_elems = (long long)L.elems; s.lcb_col_block = L.cb; s.lcb_units = L.n_units();
        s.n_workgroups += L.n_units() + ceil_div(L.n_rows(), kWavesPerWG);
        stored += (long long)L.elems;
        s.rate_fill0 = p.nnz > 0 ? (double)(stored - p.nnz) / p.nnz : 0.0;
        // values + local columns streamed, every unit's slice of x, the partials written and read
        s.data_X += (long long)L.elems * (vb + 2) + (long long)L.n_units() * std::min<long long>(L.cb, xlen) * vb + 2ll * L.n_cb * L.n_rows() * 8;
    }
    if (dev_in) { if (int rc = devpack_finish_panels(p)) return rc; }      // a device-built plan comes back uploaded: the parent's partial-result buffers too
    lap("parent upload");
    s.pre_ms = std::chrono::duration<double, std::milli>(clk::now() - t_begin).count();
    return DASP_OK;
}

int build_plan(Plan &p, const int *rp, const int *ci, const void *val, const DevCsr *dev)
{
    p.geo = geometry_for(p.precision);
    if (p.precision == 64) return build_impl<double>(p, rp, ci, static_cast<const double *>(val), dev);
    return build_impl<_Float16>(p, rp, ci, static_cast<const _Float16 *>(val), dev);
}

}  // namespace dasp

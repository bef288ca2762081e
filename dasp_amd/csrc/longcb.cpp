// longcb.cpp -- column-blocked long rows of a column-panel plan: the rule, the host packer, the checks of a loaded plan file.
// Layout and kernels: plan.hpp (struct LongCB), kernels.hip (dasp_lcb_kernel / dasp_lcb_reduce_kernel), DESIGN.md section 4.3.
#include <algorithm>
#include <atomic>
#include <cstdint>
#include <cstring>
#include <string>
#include <thread>
#include <vector>

#include "plan.hpp"

namespace dasp {

namespace {
int lcb_col_block(const Plan &p) { return p.precision == 64 ? 16384 : 32768; }      // 128 KiB (f64) / 64 KiB (f16) of LDS per workgroup
}

// rows of >= h nonzeros, h = max(block_longest, 64 per column block): a piece then averages a wave's worth of elements per step.  Auto: when those rows hold
// at least a quarter of the nonzeros (powerlaw_1M f64: 2565 rows of >= 4096 hold 68 %); 1 forces (h = block_longest), -1 turns it off.
// share_den: the automatic rule asks for >= 1 / share_den of the nonzeros in those rows: a quarter, for column panels and for the two-phase hybrid alike (r6, f16: powerlaw_1M,
// 68 % in 4375 hub rows, 198 -> 124 us as a hybrid; rmat_2M, 9 % in 232 rows, 82.0 -> 91.3 and x0.5 51.5 -> 55.7: every unit stages 64 KB of x for a few hundred elements)
// per_block: a hub row holds >= per_block nonzeros per column block on average -- 64 for column panels (the powerlaw_1M f64 sweep, profiles/r05_long_cb.md), 128 = one
// whole step of the hub kernel for the two-phase hybrid, whose alternative is the better one (rmat_2M x0.5 f16: 49 % of the nonzeros in rows of >= 2048 = 64 per block,
// 51.7 -> 55.9 us as a hybrid)
int decide_long_cb(const Plan &p, const int *rp, int P, std::vector<unsigned char> &in_lcb, int share_den, int per_block)
{
    in_lcb.clear();
    if (p.opt.long_cb < 0 || P < 2 || p.opt.n_parts > 0 || !p.dst_map.empty() || p.m <= 0) return 0;
    const int cb = lcb_col_block(p), n_cb = std::max(1, (p.n + cb - 1) / cb);
    const long long h = p.opt.long_cb > 0 ? (long long)std::max(6, p.opt.block_longest) : std::max<long long>(p.opt.block_longest, (long long)per_block * n_cb);
    long long nnz_l = 0; int rows = 0;
    for (int i = 0; i < p.m; ++i) { const int len = rp[i + 1] - rp[i]; if (len >= h) { nnz_l += len; ++rows; } }
    if (rows == 0 || (long long)rows * n_cb >= (1ll << 27)) return 0;
    if (nnz_l >= (long long)p.nnz) return 0;      // nothing would be left for the panels (every panel empty = no panel plan at all): such a matrix is the long-row kernel's
    if (p.opt.long_cb == 0 && nnz_l * share_den < (long long)p.nnz) return 0;
    in_lcb.assign((size_t)p.m, 0);
    for (int i = 0; i < p.m; ++i) if (rp[i + 1] - rp[i] >= h) in_lcb[(size_t)i] = 1;
    return rows;
}

int build_long_cb(Plan &p, const int *rp, const int *ci, const void *val, const std::vector<unsigned char> &in_lcb, const int *slot_of_row)
{
    LongCB &L = p.lcb;
    L = LongCB{};
    const int m = p.m, vb = p.geo.vbytes, A = kLcbStep;          // a piece is a whole number of steps
    L.cb = lcb_col_block(p); L.n_cb = std::max(1, (p.n + L.cb - 1) / L.cb);
    L.h = 1 << 30;
    for (int i = 0; i < m; ++i) if (in_lcb[(size_t)i]) { L.row_id.push_back(i); L.row_dst.push_back(slot_of_row ? slot_of_row[i] : i); L.h = std::min(L.h, rp[i + 1] - rp[i]); }
    const int nL = L.n_rows(), n_cb = L.n_cb, cb = L.cb;
    const int threads = resolve_threads(p.opt.host_threads);
    // elements of every piece (row-parallel), padded to A
    std::vector<int> cnt((size_t)n_cb * (size_t)nL, 0);
    auto rows_par = [&](auto f) {
        const int T = std::max(1, std::min(threads, nL));
        std::atomic<int> next{0};
        std::vector<std::thread> th;
        auto work = [&] { for (int i = next++; i < nL; i = next++) f(i); };
        for (int t = 1; t < T; ++t) th.emplace_back(work);
        work();
        for (auto &t : th) t.join();
    };
    rows_par([&](int i) { const int r = L.row_id[(size_t)i]; for (int j = rp[r]; j < rp[r + 1]; ++j) cnt[(size_t)(ci[j] / cb) * (size_t)nL + (size_t)i]++; });
    L.ptr.assign((size_t)n_cb * (size_t)nL + 1, 0);
    long long run = 0;
    for (size_t q = 0; q < cnt.size(); ++q) { L.ptr[q] = (int)run; run += (cnt[q] + A - 1) / A * A; if (run >= (1ll << 31) - 64) { set_error("long_cb: too many elements"); return DASP_ERR_ARG; } }
    L.ptr[cnt.size()] = (int)run;
    L.elems = (size_t)run;
    try { L.lcol.resize(L.elems); L.val.resize(L.elems * (size_t)vb); }
    catch (const std::bad_alloc &) { set_error("out of host memory"); return DASP_ERR_NOMEM; }
    const char *vsrc = static_cast<const char *>(val);
    rows_par([&](int i) {
        const int r = L.row_id[(size_t)i];
        std::vector<int> cur((size_t)n_cb, 0);
        for (int j = rp[r]; j < rp[r + 1]; ++j) {
            const int c = ci[j] / cb;
            const size_t e = (size_t)L.ptr[(size_t)c * (size_t)nL + (size_t)i] + (size_t)cur[(size_t)c]++;
            L.lcol[e] = (uint16_t)(ci[j] - c * cb);
            std::memcpy(L.val.data() + e * (size_t)vb, vsrc + (size_t)j * (size_t)vb, (size_t)vb);
        }
        for (int c = 0; c < n_cb; ++c) {
            const size_t q = (size_t)c * (size_t)nL + (size_t)i;
            for (size_t e = (size_t)L.ptr[q] + (size_t)cnt[q]; e < (size_t)L.ptr[q + 1]; ++e) { L.lcol[e] = kLcbPadCol; std::memset(L.val.data() + e * (size_t)vb, 0, (size_t)vb); }
        }
    });
    // units: runs of pieces of one column block with ~kLcbUnitElems elements (and at most kLcbUnitPieces pieces)
    for (int c = 0; c < n_cb; ++c) {
        int q0 = c * nL;
        const int qe = (c + 1) * nL;
        while (q0 < qe) {
            int q1 = q0 + 1;
            while (q1 < qe && q1 - q0 < kLcbUnitPieces && L.ptr[(size_t)q1] - L.ptr[(size_t)q0] < kLcbUnitElems) ++q1;
            if (L.ptr[(size_t)q1] > L.ptr[(size_t)q0]) { L.unit.push_back(c); L.unit.push_back(q0); L.unit.push_back(q1); }
            // the sums of a unit's steps are parked in LDS: a piece beyond cb elements (a row that repeats columns thousands of times) does not fit the slice reserved for them
            if ((L.ptr[(size_t)q1] - L.ptr[(size_t)q0]) / kLcbStep > kLcbUnitElems / kLcbStep + kLcbUnitPieces) { L = LongCB{}; return 1; }
            q0 = q1;
        }
    }
    return DASP_OK;
}

bool validate_long_cb(const Plan &p, int n_panels, std::string &why)
{
    auto fail = [&](const char *w) { why = w; return false; };
    const LongCB &L = p.lcb;
    const int nL = L.n_rows();
    if (nL == 0) return L.ptr.empty() && L.unit.empty() && L.lcol.empty() && L.val.empty() && L.row_id.empty() ? true : fail("long_cb arrays without rows");
    if ((n_panels < 1 && !p.two_phase) || p.opt.n_parts > 0) return fail("long_cb outside a column-panel or two-phase plan");
    const int vb = p.geo.vbytes, A = kLcbStep;
    if (L.cb < 8 || L.cb % 8 || L.cb > 65528 || L.n_cb != std::max(1, (p.n + L.cb - 1) / L.cb)) return fail("long_cb column blocks");
    // the kernel's dynamic LDS: the block's slice of x + 16 + one sum per step and piece of a unit (kernels.hip launch_spmv) -- all of it inside the 160 KiB the attribute allows
    if ((size_t)L.cb * (size_t)vb + 16 + (size_t)(kLcbUnitElems / kLcbStep + kLcbUnitPieces) * 8 > 160 * 1024) return fail("long_cb column block beyond the LDS");
    if (L.row_id.size() != (size_t)nL || L.ptr.size() != (size_t)L.n_cb * (size_t)nL + 1 || L.ptr[0] != 0) return fail("long_cb tables");
    if ((size_t)L.ptr.back() != L.elems || L.lcol.size() != L.elems || L.val.size() != L.elems * (size_t)vb) return fail("long_cb streams");
    std::vector<bool> seen((size_t)p.m, false);
    for (int i = 0; i < nL; ++i) {
        const int d = L.row_dst[(size_t)i], r = L.row_id[(size_t)i];
        if ((unsigned)d >= (unsigned)p.m || (unsigned)r >= (unsigned)p.m || seen[(size_t)d]) return fail("long_cb destination out of range or used twice");
        seen[(size_t)d] = true;
        if ((p.opt.y_order == DASP_Y_NATURAL ? r : p.order[(size_t)d]) != r) return fail("long_cb destination is not the row's position");
    }
    for (size_t q = 0; q + 1 < L.ptr.size(); ++q) {
        if (L.ptr[q + 1] < L.ptr[q] || L.ptr[q] % A) return fail("long_cb piece offsets");
        const int c = (int)(q / (size_t)nL), width = std::min(L.cb, p.n - c * L.cb);
        for (int e = L.ptr[q]; e < L.ptr[q + 1]; ++e) if (L.lcol[(size_t)e] != kLcbPadCol && (int)L.lcol[(size_t)e] >= width) return fail("long_cb local column beyond its block");
    }
    if (L.unit.size() % 3) return fail("long_cb unit table");
    std::vector<bool> covered((size_t)L.n_cb * (size_t)nL, false);
    for (int u = 0; u < L.n_units(); ++u) {
        const int c = L.unit[3 * (size_t)u], q0 = L.unit[3 * (size_t)u + 1], q1 = L.unit[3 * (size_t)u + 2];
        if (c < 0 || c >= L.n_cb || q0 < c * nL || q1 > (c + 1) * nL || q1 <= q0 || q1 - q0 > kLcbUnitPieces) return fail("long_cb unit range");
        if ((L.ptr[(size_t)q1] - L.ptr[(size_t)q0]) / kLcbStep > kLcbUnitElems / kLcbStep + kLcbUnitPieces) return fail("long_cb unit with more steps than its LDS slice holds");
        for (int q = q0; q < q1; ++q) { if (covered[(size_t)q]) return fail("long_cb piece in two units"); covered[(size_t)q] = true; }
    }
    for (size_t q = 0; q < covered.size(); ++q) if (!covered[q] && L.ptr[q + 1] > L.ptr[q]) return fail("long_cb piece in no unit");
    if (p.stats.lcb_rows != nL || p.stats.lcb_elems != (long long)L.elems || p.stats.lcb_col_block != L.cb || p.stats.lcb_units != L.n_units()) return fail("long_cb counters");
    return true;      // (that the panels do not hold these rows too: planio.cpp read_plan adds the nonzeros up)
}

}  // namespace dasp

// gen.cpp -- seeded synthetic stand-ins for the SuiteSparse matrices BASELINE.json names.
// Neither the build container nor the GPU box has .mtx files or a network, so the benchmark
// inputs are generated: each named stand-in follows the collection's dimensions, row-length
// statistics and structure class.  Row r of a matrix depends only on
// (name, scale, r), so any row range can be produced independently (multi-GPU ranks build
// only their slice) and reproducibly.
//
//   cop20k_A      121192 rows, symmetric, avg 21.7 / max ~81 per row, empty rows, +-4000 band
//   nlpkkt160     8345600 = 2 dof x 160x160x163 grid, symmetric, <= 28 per row (19-pt + 9-pt coupling)
//   Queen_4147    4147110 rows, 3 dof x 27-point stencil (<= 81 per row), symmetric
//   HV15R         2017169 rows, 5 dof x 27-point stencil (135) with 2% / 0.2% extended rows (375 / 484)
//   HV15R-unstructured  the same dimensions and row lengths, but what an UNSTRUCTURED mesh leaves after a bandwidth-reducing
//                 ordering: every node couples with itself and 26 / 74 / 95 nodes scattered over a band of +-6000 nodes (one draw per
//                 stratum of the band, so a row is 27 / 75 / 96 runs of 5 adjacent columns at scattered positions) -- no grid
//                 numbering, no plane structure, neighbouring rows share few columns.  Shows how much of the HV15R stand-in's
//                 roofline fraction is owed to its structured-grid numbering.
//   webbase-1M    1000005 rows, power-law lengths (mean ~3.1, max 4700); HOST-BLOCK columns: the pages of a host are contiguous
//                 (a crawl in URL order), host sizes are power-law distributed, 85 % of a page's links stay inside its host
//                 (half of them within +-32 pages, half skewed to the host's first pages), 3 % go to a neighbouring host, 12 %
//                 to globally popular pages (power-law popularity, the popular ids scattered over the whole range)
//   ljournal-2008 5363260 rows, power-law lengths (mean ~14.7, max 2469); COMMUNITY columns: contiguous communities of
//                 power-law size, 45 % of a user's links inside the community, 15 % to the four neighbouring ones, 40 % to
//                 globally popular users (scattered ids).  Assumed, not measured: no SuiteSparse file is available here;
//                 the figures follow what is commonly reported for host-ordered web crawls and LiveJournal communities.
//   webbase-1M-uniform / ljournal-2008-uniform: the round-1 stand-ins (70 % near / 30 % uniform; all uniform): worst-case gathers
//   powerlaw_1M   2^20 rows, Zipf(1.8) lengths clipped at 200000 (mean ~48), uniform columns
//   rmat_2M       2^21 rows, R-MAT / Graph500 (a, b, c, d = 0.57, 0.19, 0.19, 0.05), 16 edges per row on average: skewed degrees AND
//                 skewed, community-structured columns -- a closer proxy for web / social graphs than uniform columns
#include <algorithm>
#include <cmath>
#include <cstring>
#include <string>
#include <thread>
#include <vector>

#include "plan.hpp"

namespace dasp {
namespace {

inline uint64_t mix(uint64_t z)
{
    z += 0x9E3779B97F4A7C15ull;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}
inline uint64_t h2(uint64_t seed, uint64_t a, uint64_t b) { return mix(mix(seed ^ mix(a)) ^ (b * 0xD6E8FEB86659FD93ull)); }
inline double u01(uint64_t h) { return ((h >> 11) + 0.5) * (1.0 / 9007199254740992.0); }

struct Synth {
    enum Kind { GRID, BAND, POWER, RMAT, BLOCKS, UNSTR } kind;
    int rows = 0, cols = 0;
    uint64_t seed = 0;
    // GRID
    int nx = 0, ny = 0, nz = 0, dof = 1;
    int variant = 0;   // 0 = full 27-pt coupling, 1 = nlpkkt (19-pt same dof, 9-pt cross), 2 = HV15R extended rows
    // BAND
    int band = 0; double deg_mean = 0, deg_sd = 0; int deg_max = 0; double tbar = 1;
    // POWER
    double alpha = 2, xmin = 1, p_zero = 0, near_frac = 0; int len_max = 0, near_w = 0;
    // RMAT
    int levels = 0; double edge_factor = 16;
    // BLOCKS (hosts / communities): contiguous blocks of power-law size; row lengths as POWER
    std::vector<int> bstart;             // [nblocks+1]
    double p_in = 0, p_nb = 0, in_near = 0, in_skew = 1, g_skew = 2; int nb_reach = 1; uint64_t perm_mul = 1;
    const char *desc = "";
};

int scaled(int full, double s) { return std::max(64, (int)std::llround(full * s)); }

bool make(const char *name, double scale, Synth &g)
{
    const std::string n(name ? name : "");
    if (!(scale > 0)) return false;
    const double c = std::cbrt(scale);
    auto grid = [&](int nx, int ny, int nz, int dof, int rows_full, int variant, uint64_t seed) {
        g.kind = Synth::GRID; g.dof = dof; g.variant = variant; g.seed = seed;
        g.nx = std::max(4, (int)std::llround(nx * c)); g.ny = std::max(4, (int)std::llround(ny * c));
        g.nz = std::max(4, (int)std::llround(nz * c));
        const long long cap = (long long)g.nx * g.ny * g.nz * dof;
        g.rows = g.cols = (int)std::min<long long>(cap, scale == 1.0 ? rows_full : (long long)scaled(rows_full, scale));
    };
    if (n == "Queen_4147") { grid(113, 111, 111, 3, 4147110, 0, 20007); g.desc = "3 dof x 27-point stencil on a 113x111x111 grid"; return true; }
    if (n == "HV15R") { grid(74, 74, 74, 5, 2017169, 2, 20006); g.desc = "5 dof x 27-point stencil on a 74^3 grid, 2% / 0.2% extended rows (375 / 484)"; return true; }
    if (n == "HV15R-unstructured") {
        grid(74, 74, 74, 5, 2017169, 2, 20016);
        g.kind = Synth::UNSTR; g.band = 6000;
        g.desc = "5 dof per node, 27 / 75 (2%) / 96 (0.2%) coupled nodes drawn one per stratum of a +-6000-node band (unstructured mesh after RCM): HV15R's size and row lengths without grid numbering";
        return true;
    }
    if (n == "nlpkkt160") { grid(160, 160, 163, 2, 8345600, 1, 20002); g.desc = "2 dof on a 160x160x163 grid, 19-point + 9-point coupling"; return true; }
    if (n == "cop20k_A") {
        g.kind = Synth::BAND; g.seed = 20001; g.rows = g.cols = scaled(121192, scale);
        g.band = std::min(4000, std::max(8, g.rows / 4)); g.deg_mean = 21.7; g.deg_sd = 14; g.deg_max = 81;
        // E[clip(N(21.7,14),0,81)] ~ 21.95 ; t_i / sqrt(tbar) propensities give E[deg_i] ~ t_i
        g.tbar = 21.95;
        g.desc = "symmetric, degree ~ clip(N(21.7,14),0,81), columns uniform in a +-4000 band";
        return true;
    }
    auto power = [&](int rows_full, double alpha, double xmin, double pz, int lmax, double nearf, int nearw, uint64_t seed) {
        g.kind = Synth::POWER; g.seed = seed; g.rows = g.cols = scaled(rows_full, scale);
        g.alpha = alpha; g.xmin = xmin; g.p_zero = pz; g.len_max = std::min(lmax, std::max(8, g.cols / 2));
        g.near_frac = nearf; g.near_w = nearw;
    };
    // contiguous blocks of Pareto(balpha) size in [bmin, bmax], laid end to end until the rows are covered
    auto blocks = [&](double balpha, int bmin, int bmax) {
        g.kind = Synth::BLOCKS;
        bmax = std::max(bmin, std::min(bmax, g.rows / 4));
        g.bstart.assign(1, 0);
        for (uint64_t k = 0; g.bstart.back() < g.rows; ++k) {
            const double v = u01(h2(g.seed ^ 0xB10Cull, k, 5));
            const int sz = (int)std::min<double>(bmax, std::floor(bmin * std::pow(v, -1.0 / (balpha - 1.0))));
            g.bstart.push_back((int)std::min<long long>(g.rows, (long long)g.bstart.back() + std::max(1, sz)));
        }
        // bijection r -> (r * perm_mul) mod rows scatters the popular ids (perm_mul odd prime, rows never a multiple of it)
        g.perm_mul = 2654435761ull;
        while (g.rows % g.perm_mul == 0) g.perm_mul += 2;
    };
    if (n == "webbase-1M-uniform") { power(1000005, 2.45, 1.32, 0.02, 4700, 0.7, 1000, 20004); g.desc = "power-law lengths; 70% of the columns within +-1000 of the row, 30% uniform (round-1 stand-in)"; return true; }
    if (n == "ljournal-2008-uniform") { power(5363260, 2.35, 4.6, 0.01, 2469, 0.0, 0, 20005); g.desc = "power-law lengths; uniform random columns (round-1 stand-in, worst-case gathers)"; return true; }
    if (n == "webbase-1M") {
        power(1000005, 2.45, 1.32, 0.02, 4700, 0.0, 0, 20004);
        blocks(1.9, 24, 60000);
        g.p_in = 0.85; g.in_near = 0.5; g.in_skew = 2.0; g.p_nb = 0.03; g.nb_reach = 1; g.g_skew = 3.0;
        g.desc = "power-law lengths; host blocks (Pareto 1.9, 24..60000 pages): 85% of the links intra-host (half +-32, half skewed to the host's first pages), 3% next host, 12% popular pages (scattered ids)";
        return true;
    }
    if (n == "ljournal-2008") {
        power(5363260, 2.35, 4.6, 0.01, 2469, 0.0, 0, 20005);
        blocks(2.2, 256, 120000);
        g.p_in = 0.45; g.in_near = 0.0; g.in_skew = 1.0; g.p_nb = 0.15; g.nb_reach = 4; g.g_skew = 2.5;
        g.desc = "power-law lengths; communities (Pareto 2.2, 256..120000 users): 45% of the links inside the community, 15% to the 4 neighbouring ones, 40% to popular users (scattered ids)";
        return true;
    }
    if (n == "powerlaw_1M") { power(1 << 20, 1.8, 2.05, 0.0, 200000, 0.0, 0, 20003); g.desc = "Zipf(1.8) lengths clipped at 200000; uniform random columns"; return true; }
    if (n == "rmat_2M") {
        g.kind = Synth::RMAT; g.seed = 20008; g.rows = g.cols = scaled(1 << 21, scale);
        g.levels = 1; while ((1ll << g.levels) < g.rows) g.levels++;
        g.edge_factor = 16; g.len_max = std::max(8, g.cols / 2);
        g.desc = "R-MAT (0.57, 0.19, 0.19, 0.05), 16 edges per row";
        return true;
    }
    return false;
}

// ---- GRID rows ------------------------------------------------------------------------
// neighbourhood class of a node for the HV15R stand-in: 0 = 27-pt, 1 = 5x5x3 (75 nodes), 2 = 5x5x4 (100 nodes)
inline int hv_class(const Synth &g, long long node)
{
    const uint64_t h = h2(g.seed, (uint64_t)node, 77) % 1000;
    return h < 2 ? 2 : (h < 22 ? 1 : 0);
}

int grid_row(const Synth &g, int row, int *out)
{
    const int dof = g.dof;
    const long long node = row / dof;
    const int da = row % dof;
    const int x = (int)(node % g.nx), y = (int)((node / g.nx) % g.ny), z = (int)(node / ((long long)g.nx * g.ny));
    int rx = 1, ry = 1, rz0 = -1, rz1 = 1, cap = 1 << 30;
    if (g.variant == 2) {
        const int cls = hv_class(g, node);
        if (cls >= 1) { rx = 2; ry = 2; }
        if (cls == 2) { rz1 = 2; cap = 484; }
    } else if (g.variant == 1) {
        rx = 2;   // the +-2x cross-dof coupling
    }
    int len = 0;
    for (int dz = rz0; dz <= rz1; ++dz) {
        const int zz = z + dz;
        if (zz < 0 || zz >= g.nz) continue;
        for (int dy = -ry; dy <= ry; ++dy) {
            const int yy = y + dy;
            if (yy < 0 || yy >= g.ny) continue;
            for (int dx = -rx; dx <= rx; ++dx) {
                const int xx = x + dx;
                if (xx < 0 || xx >= g.nx) continue;
                const long long nb = ((long long)zz * g.ny + yy) * g.nx + xx;
                for (int db = 0; db < dof; ++db) {
                    if (g.variant == 1) {
                        const int nzc = (dx != 0) + (dy != 0) + (dz != 0);
                        const int l1 = std::abs(dx) + std::abs(dy) + std::abs(dz);
                        bool ok;
                        if (da == db) ok = std::abs(dx) <= 1 && nzc <= 2;                 // 19-point
                        else ok = l1 <= 1 || (std::abs(dx) == 2 && dy == 0 && dz == 0);    // 7-point + (+-2,0,0)
                        if (!ok) continue;
                    }
                    const long long col = nb * dof + db;
                    if (col >= g.cols) continue;
                    if (len >= cap) continue;
                    if (out) out[len] = (int)col;
                    ++len;
                }
            }
        }
    }
    return len;
}

// ---- UNSTR rows (HV15R-unstructured): node `row / dof` couples with `deg` nodes, one per stratum of its band window; the stratum
// holding the node itself yields the node, so the diagonal block is always present and a row has no duplicates and comes out sorted
int unstr_row(const Synth &g, int row, int *out)
{
    const int dof = g.dof;
    const long long nodes = ((long long)g.cols + dof - 1) / dof, node = row / dof;
    const int cls = hv_class(g, node);
    const long long W = std::min<long long>(nodes, 2ll * g.band + 1);
    const int deg = (int)std::min<long long>(W, cls == 2 ? 96 : cls == 1 ? 75 : 27);
    const long long start = std::min(std::max(0ll, node - g.band), nodes - W);
    int len = 0;
    for (int k = 0; k < deg; ++k) {
        const long long lo = start + W * k / deg, hi = start + W * (k + 1) / deg;       // stratum k: [lo, hi), never empty (W >= deg)
        long long nb = lo + (long long)(h2(g.seed, (uint64_t)node, (uint64_t)k + 1000) % (uint64_t)(hi - lo));
        if (node >= lo && node < hi) nb = node;
        for (int db = 0; db < dof; ++db) {
            const long long col = nb * dof + db;
            if (col >= g.cols) continue;
            if (out) out[len] = (int)col;
            ++len;
        }
    }
    return len;
}

// ---- BAND rows (cop20k_A stand-in): symmetric by construction -------------------------
inline double band_target(const Synth &g, int i)
{
    // Box-Muller from two hashes of the row
    const double u1 = u01(h2(g.seed, (uint64_t)i, 1)), u2 = u01(h2(g.seed, (uint64_t)i, 2));
    const double nrm = std::sqrt(-2.0 * std::log(u1)) * std::cos(6.283185307179586 * u2);
    double t = g.deg_mean + g.deg_sd * nrm;
    return std::min((double)g.deg_max, std::max(0.0, t));
}

// tg[j - tg0] caches band_target(j) for the rows a call can touch (the Box-Muller draw is the
// expensive part; the window test itself is one hash per candidate)
int band_row(const Synth &g, int i, int *out, const float *tg, int tg0)
{
    const double ti = tg[i - tg0];
    if (ti <= 0) return 0;
    const int lo = std::max(0, i - g.band), hi = std::min(g.rows - 1, i + g.band);
    // window actually available to row i (rows near the ends see a one-sided band)
    const double si = ti / std::sqrt(g.tbar);
    int len = 0;
    for (int j = lo; j <= hi; ++j) {
        bool take;
        if (j == i) take = true;
        else {
            const double tj = tg[j - tg0];
            if (tj <= 0) continue;
            const double p = si * (tj / std::sqrt(g.tbar)) / (2.0 * g.band);
            const int a = std::max(i, j), b = std::min(i, j);
            take = u01(h2(g.seed ^ 0xABCDEFull, (uint64_t)a, (uint64_t)b)) < p;
        }
        if (take) { if (out) out[len] = j; ++len; }
    }
    return len;
}

// ---- POWER rows ------------------------------------------------------------------------
inline int power_len(const Synth &g, int i)
{
    const double u = u01(h2(g.seed, (uint64_t)i, 11));
    if (u < g.p_zero) return 0;
    const double v = u01(h2(g.seed, (uint64_t)i, 12));
    const double xlen = g.xmin * std::pow(v, -1.0 / (g.alpha - 1.0));
    return (int)std::min<double>(g.len_max, std::floor(xlen));
}

int power_row(const Synth &g, int i, int *out)
{
    const int len = power_len(g, i);
    if (!out) return len;
    for (int k = 0; k < len; ++k) {
        const uint64_t h = h2(g.seed ^ 0x5151ull, (uint64_t)i, (uint64_t)k);
        int col;
        if (u01(h) < g.near_frac) {
            const int w = std::min(g.near_w, g.cols / 2);
            col = i + (int)(mix(h) % (uint64_t)(2 * w + 1)) - w;
            if (col < 0) col += g.cols;
            if (col >= g.cols) col -= g.cols;
        } else col = (int)(mix(h ^ 0x77ull) % (uint64_t)g.cols);
        out[k] = col;
    }
    return len;
}

// ---- BLOCKS rows (host / community structure) -----------------------------------------
int blocks_row(const Synth &g, int i, int *out)
{
    const int len = power_len(g, i);
    if (!out) return len;
    const int nb = (int)g.bstart.size() - 1;
    const int b = (int)(std::upper_bound(g.bstart.begin(), g.bstart.end(), i) - g.bstart.begin()) - 1;
    const int b0 = g.bstart[(size_t)b], bs = g.bstart[(size_t)b + 1] - b0;
    for (int k = 0; k < len; ++k) {
        const uint64_t h = h2(g.seed ^ 0x5151ull, (uint64_t)i, (uint64_t)k);
        const double u = u01(h), v = u01(mix(h ^ 0x77ull));
        long long col;
        if (u < g.p_in) {
            if (u < g.p_in * g.in_near) {                       // a page next to this one
                col = (long long)i + (long long)(mix(h ^ 0x99ull) % 65ull) - 32;
                col = std::min<long long>(b0 + bs - 1, std::max<long long>(b0, col));
            } else col = b0 + (long long)(bs * std::pow(v, g.in_skew));
        } else if (u < g.p_in + g.p_nb) {                       // a neighbouring block
            int d = 1 + (int)(mix(h ^ 0x33ull) % (uint64_t)g.nb_reach);
            if (mix(h ^ 0x44ull) & 1) d = -d;
            const int bb = std::min(nb - 1, std::max(0, b + d));
            const int c0 = g.bstart[(size_t)bb], cs = g.bstart[(size_t)bb + 1] - c0;
            col = c0 + (long long)(cs * v);
        } else {                                                // a globally popular target, ids scattered by a bijection
            const long long r = (long long)(g.cols * std::pow(v, g.g_skew));
            col = (long long)(((unsigned __int128)(uint64_t)r * g.perm_mul) % (uint64_t)g.cols);
        }
        out[k] = (int)std::min<long long>(g.cols - 1, std::max<long long>(0, col));
    }
    return len;
}

// ---- RMAT rows --------------------------------------------------------------------------
// Row i of an R-MAT matrix, generated on its own: the expected degree of source i is edges * prod over its bits of
// (a+b = 0.76 for a 0 bit, c+d = 0.24 for a 1 bit); each destination bit is 1 with probability b/(a+b) = 1/4 under a 0 source
// bit and d/(c+d) ~ 7/32 under a 1 source bit.  Entries are not de-duplicated (the plan takes duplicates as they come).
inline int rmat_len(const Synth &g, int i)
{
    double d = g.edge_factor * (double)g.rows;
    for (int b = 0; b < g.levels; ++b) d *= ((i >> b) & 1) ? 0.24 : 0.76;
    const double u = u01(h2(g.seed, (uint64_t)i, 21));
    return (int)std::min<double>(g.len_max, std::floor(d + u));      // stochastic rounding: E[len] = d
}

int rmat_row(const Synth &g, int i, int *out)
{
    const int len = rmat_len(g, i);
    if (!out) return len;
    for (int k = 0; k < len; ++k) {
        uint64_t h = h2(g.seed ^ 0x7a7aull, (uint64_t)i, (uint64_t)k);
        long long col = 0;
        for (int b = 0, used = 0; b < g.levels; ++b, used += 5) {
            if (used + 5 > 64) { h = mix(h ^ 0x1234567ull); used = 0; }
            const unsigned r5 = (unsigned)(h >> used) & 31u;
            const unsigned cut = ((i >> b) & 1) ? 7u : 8u;           // P(bit = 1) = 7/32 or 8/32
            if (r5 < cut) col |= 1ll << b;
        }
        out[k] = (int)(col % g.cols);
    }
    return len;
}

// the graph generators draw a row's columns one by one: they come out in random order.  A SuiteSparse .mtx lists its entries column by column, and mmio_allinone's
// COO -> CSR keeps the file order inside a row (src/mmio_highlevel.h), so a row of a general matrix (webbase-1M, ljournal-2008: directed graphs) arrives with ascending
// columns -- r4: the stand-ins do too (until then powerlaw_1M ran 16 %, rmat_2M 13 %, ljournal-2008 3 % slower than the same kernels on the sorted rows:
// tools/hot_cols_probe.py).  Duplicate columns stay separate entries (the plan takes them as they come), nnz is unchanged.
inline int any_row(const Synth &g, int row, int *out, const float *tg, int tg0)
{
    int len;
    switch (g.kind) {
        case Synth::GRID: return grid_row(g, row, out);
        case Synth::UNSTR: return unstr_row(g, row, out);
        case Synth::BAND: return band_row(g, row, out, tg, tg0);
        case Synth::RMAT: len = rmat_row(g, row, out); break;
        case Synth::BLOCKS: len = blocks_row(g, row, out); break;
        default: len = power_row(g, row, out); break;
    }
    if (out && len > 1) std::sort(out, out + len);
    return len;
}

// per-call cache of band targets for rows [r0 - band, r1 + band)
void band_cache(const Synth &g, int r0, int r1, std::vector<float> &tg, int &tg0)
{
    tg0 = 0;
    if (g.kind != Synth::BAND) return;
    const int lo = std::max(0, r0 - g.band), hi = std::min(g.rows, r1 + g.band);
    tg0 = lo;
    tg.resize((size_t)std::max(0, hi - lo));
    for (int j = lo; j < hi; ++j) tg[(size_t)(j - lo)] = (float)band_target(g, j);
}

template <class F>
void par_rows(int r0, int r1, F f)
{
    const int nt = resolve_threads(0);
    const long long n = (long long)r1 - r0;
    if (n <= 0) return;
    const int parts = (int)std::min<long long>(nt, std::max<long long>(1, n / 256));
    std::vector<std::thread> th;
    for (int t = 0; t < parts; ++t) {
        const int b = r0 + (int)(n * t / parts), e = r0 + (int)(n * (t + 1) / parts);
        th.emplace_back([=] { for (int r = b; r < e; ++r) f(r); });
    }
    for (auto &x : th) x.join();
}

}  // namespace
}  // namespace dasp

using namespace dasp;

template <class F>
static int synth_guard(F f) noexcept
{
    try { return f(); }
    catch (const std::bad_alloc &) { set_error("synthetic generator: out of host memory"); return DASP_ERR_NOMEM; }
    catch (const std::exception &e) { set_error(std::string("synthetic generator: ") + e.what()); return DASP_ERR_ARG; }
}

extern "C" int dasp_synth_dims(const char *name, double scale, int *rows, int *cols)
{
    return synth_guard([&] {
        Synth g;
        if (!make(name, scale, g) || !rows || !cols) { set_error("unknown synthetic matrix name"); return (int)DASP_ERR_ARG; }
        *rows = g.rows; *cols = g.cols;
        return (int)DASP_OK;
    });
}

extern "C" const char *dasp_synth_generator(const char *name)
{
    static thread_local std::string text;
    Synth g;
    if (!make(name, 1.0, g)) { set_error("unknown synthetic matrix name"); return nullptr; }
    text = std::string("dasp_amd/csrc/gen.cpp seed ") + std::to_string((unsigned long long)g.seed) + ": " + g.desc;
    if (g.kind != Synth::GRID && g.kind != Synth::UNSTR && g.kind != Synth::BAND) text += "; columns ascending inside a row (as mmio_allinone reads a general .mtx)";
    return text.c_str();
}

extern "C" int dasp_synth_row_lengths(const char *name, double scale, int row_begin, int row_end, int *len_out)
{
    return synth_guard([&] {
        Synth g;
        if (!make(name, scale, g) || !len_out || row_begin < 0 || row_end > g.rows || row_begin > row_end) {
            set_error("bad arguments to dasp_synth_row_lengths"); return (int)DASP_ERR_ARG;
        }
        std::vector<float> tg; int tg0;
        band_cache(g, row_begin, row_end, tg, tg0);
        par_rows(row_begin, row_end, [&](int r) { len_out[r - row_begin] = any_row(g, r, nullptr, tg.data(), tg0); });
        return (int)DASP_OK;
    });
}

extern "C" int dasp_synth_rows(const char *name, double scale, int row_begin, int row_end, const int *rp, int *col_idx_out)
{
    return synth_guard([&] {
        Synth g;
        if (!make(name, scale, g) || !rp || !col_idx_out || row_begin < 0 || row_end > g.rows || row_begin > row_end) {
            set_error("bad arguments to dasp_synth_rows"); return (int)DASP_ERR_ARG;
        }
        std::vector<float> tg; int tg0;
        band_cache(g, row_begin, row_end, tg, tg0);
        par_rows(row_begin, row_end, [&](int r) { any_row(g, r, col_idx_out + rp[r - row_begin], tg.data(), tg0); });
        return (int)DASP_OK;
    });
}

// tests/san/san_driver.cpp -- TEST INFRASTRUCTURE: drives the HOST side of libdasp_amd (loader, CSR cache, classifier + packers with every
// layout option, serialised plans, the synthetic generators, the multi-GPU host split, the persistent worker pool from several caller
// threads) through the C ABI under AddressSanitizer + UBSan / ThreadSanitizer.  usage: san_driver <dir with *.mtx> <scratch dir>
#include <atomic>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <dirent.h>
#include <string>
#include <thread>
#include <vector>

#include "../../include/dasp_amd.h"

static int fails = 0;
#define CHECK(c) do { if (!(c)) { std::fprintf(stderr, "CHECK failed %s:%d: %s (%s)\n", __FILE__, __LINE__, #c, dasp_last_error()); ++fails; } } while (0)

struct Csr { int m = 0, n = 0, nnz = 0; std::vector<int> rp, ci; std::vector<double> v64; std::vector<uint16_t> v16; };

static Csr synth(const char *name, double scale)
{
    Csr c;
    CHECK(dasp_synth_dims(name, scale, &c.m, &c.n) == 0);
    std::vector<int> len((size_t)c.m);
    CHECK(dasp_synth_row_lengths(name, scale, 0, c.m, len.data()) == 0);
    c.rp.assign((size_t)c.m + 1, 0);
    for (int i = 0; i < c.m; ++i) c.rp[(size_t)i + 1] = c.rp[(size_t)i] + len[(size_t)i];
    c.nnz = c.rp[(size_t)c.m];
    c.ci.resize((size_t)c.nnz + 1);
    CHECK(dasp_synth_rows(name, scale, 0, c.m, c.rp.data(), c.ci.data()) == 0);
    c.v64.assign((size_t)c.nnz + 1, 1.0);
    c.v16.assign((size_t)c.nnz + 1, 0x3C00);
    return c;
}

static void plans_of(const Csr &c, const std::string &scratch, int tag)
{
    struct O { int x_window, row_window, cid16, col_panels, slab, hybrid, piece, pairs, cid8, natural; double thr; int longest; int two_phase = 0, long_cb = 0, tp_cb = 0, tp_rb = 0; };
    const O opts[] = {
        {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0.75, 256}, {81920, 128, 1, 1, 4, -1, -1, 0, 0, 0, 0.75, 256}, {-1, 0, 1, 1, 4, -1, -1, 2, 0, 1, 0.5, 64},
        {-1, 0, -1, 3, 12, -1, 40, -1, -1, 0, 1.0, 256}, {163840, 256, 1, 1, 4, 1, -1, 0, 0, 1, 0.25, 128}, {-1, 0, 1, 1, 32, -1, 5, 1, 0, 0, 0.75, 1000000},
        // r5: the two-phase form (f16 only: applied to the f16 plan of the pair), small blocks so that every matrix has many tiles; column panels with column-blocked long rows
        {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0.75, 256, 1, 0, 64, 16}, {0, 0, 0, 0, 0, 0, 0, 0, 0, 1, 0.75, 256, 1, 0, 0, 0}, {-1, 0, 0, 3, 0, -1, 0, 0, 0, 0, 0.75, 64, -1, 1, 0, 0}, {-1, 0, 0, 2, 0, -1, 0, 0, 0, 1, 0.75, 256, -1, 1, 0, 0},
        // r6: the f16 hybrid (hub rows of a two-phase plan column-blocked: forced, rows of >= 64), both y orders; window heights that are not a multiple of 64 (whole blocks)
        {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0.75, 64, 1, 1, 256, 64}, {0, 0, 0, 0, 0, 0, 0, 0, 0, 1, 0.75, 64, 1, 1, 0, 0}, {81920, 432, 1, 1, 4, -1, -1, 0, 0, 0, 0.75, 256}, {81920, 80, 0, 1, 4, -1, -1, 0, 0, 1, 0.75, 256},
    };
    int k = 0;
    for (const O &o : opts) {
        for (int prec : {64, 16}) {
            dasp_options_t opt;
            dasp_options_default(&opt);
            opt.x_window = o.x_window; opt.row_window = o.row_window; opt.cid16 = o.cid16; opt.col_panels = o.col_panels; opt.slab_max_len = o.slab;
            opt.x_window_hybrid = o.hybrid; opt.piece_min_len = o.piece; opt.chunk_pairs = o.pairs; opt.cid8 = o.cid8; opt.y_order = o.natural;
            opt.threshold = o.thr; opt.block_longest = o.longest; opt.host_threads = 1 + (k % 5);
            opt.sort_columns = k % 3 == 0 ? 1 : 0;
            opt.two_phase = o.two_phase > 0 && prec != 16 ? 0 : o.two_phase; opt.tp_col_block = o.tp_cb; opt.tp_row_block = o.tp_rb; opt.long_cb = o.long_cb;
            dasp_plan_t *p = nullptr;
            const void *val = prec == 64 ? (const void *)c.v64.data() : (const void *)c.v16.data();
            const int rc = dasp_plan_create(&p, prec, c.m, c.n, c.nnz, c.rp.data(), c.ci.data(), val, &opt);
            CHECK(rc == 0);
            if (rc) continue;
            dasp_stats_t st;
            CHECK(dasp_plan_stats(p, &st) == 0 && st.nnzA == c.nnz);
            CHECK(st.nnz_short + st.nnz_long + st.origin_nnz_reg + st.nnz_irreg == c.nnz);
            const int *ord = dasp_plan_order(p);
            long long sum = 0;
            for (int i = 0; i < c.m; ++i) sum += ord[i];
            CHECK(sum == (long long)c.m * (c.m - 1) / 2);
            const std::string path = scratch + "/plan_" + std::to_string(tag) + "_" + std::to_string(k) + ".bin";
            CHECK(dasp_plan_save(p, path.c_str()) == 0);
            dasp_plan_t *q = nullptr;
            CHECK(dasp_plan_load(&q, path.c_str()) == 0);
            if (q) { dasp_stats_t s2; CHECK(dasp_plan_stats(q, &s2) == 0 && s2.fill0_nnz_reg == st.fill0_nnz_reg); dasp_plan_destroy(q); }
            std::remove(path.c_str());
            CHECK(dasp_plan_upload(p) == DASP_ERR_NO_DEVICE);
            dasp_plan_destroy(p);
            ++k;
        }
    }
}

int main(int argc, char **argv)
{
    if (argc < 3) { std::fprintf(stderr, "usage: san_driver <mtx dir> <scratch dir>\n"); return 2; }
    const std::string dir = argv[1], scratch = argv[2];
    // 1. every fixture through the loader (f64 and f16), the CSR cache, and the packers
    std::vector<std::string> files;
    if (DIR *d = opendir(dir.c_str())) {
        while (dirent *e = readdir(d)) { const std::string f = e->d_name; if (f.size() > 4 && f.substr(f.size() - 4) == ".mtx") files.push_back(dir + "/" + f); }
        closedir(d);
    }
    CHECK(!files.empty());
    int tag = 0;
    for (const std::string &f : files) {
        int m, n, nnz, sym, *rp = nullptr, *ci = nullptr; double *v = nullptr; uint16_t *h = nullptr;
        const int rc = dasp_mmio_allinone_f64(&m, &n, &nnz, &sym, &rp, &ci, &v, f.c_str());
        if (rc != 0) continue;                      // the malformed fixtures: the point is that they fail cleanly
        int m2, n2, nnz2, sym2, *rp2 = nullptr, *ci2 = nullptr;
        CHECK(dasp_mmio_allinone_f16(&m2, &n2, &nnz2, &sym2, &rp2, &ci2, &h, f.c_str()) == 0 && nnz2 == nnz);
        const std::string cache = scratch + "/csr_" + std::to_string(tag) + ".bin";
        CHECK(dasp_csr_save(cache.c_str(), 64, m, n, nnz, sym, rp, ci, v) == 0);
        int m3, n3, nnz3, sym3, *rp3 = nullptr, *ci3 = nullptr; void *v3 = nullptr;
        CHECK(dasp_csr_load(cache.c_str(), 64, &m3, &n3, &nnz3, &sym3, &rp3, &ci3, &v3) == 0 && nnz3 == nnz && std::memcmp(ci3, ci, sizeof(int) * (size_t)nnz) == 0);
        std::remove(cache.c_str());
        Csr c; c.m = m; c.n = n; c.nnz = nnz; c.rp.assign(rp, rp + m + 1); c.ci.assign(ci, ci + nnz); c.ci.push_back(0);
        c.v64.assign(v, v + nnz); c.v64.push_back(0); c.v16.assign(h, h + nnz); c.v16.push_back(0);
        plans_of(c, scratch, tag++);
        for (void *p : {(void *)rp, (void *)ci, (void *)v, (void *)rp2, (void *)ci2, (void *)h, (void *)rp3, (void *)ci3, v3}) dasp_free(p);
    }
    // 2. the synthetic generators at test scale, every layout option
    const char *names[] = {"cop20k_A", "nlpkkt160", "powerlaw_1M", "webbase-1M", "ljournal-2008", "HV15R", "Queen_4147", "rmat_2M", "HV15R-unstructured"};
    const double scales[] = {0.2, 0.004, 0.02, 0.03, 0.005, 0.01, 0.005, 0.01, 0.01};
    std::vector<Csr> mats;
    for (size_t i = 0; i < sizeof names / sizeof *names; ++i) mats.push_back(synth(names[i], scales[i]));
    for (const Csr &c : mats) plans_of(c, scratch, tag++);
    // 3. the worker pool under several caller threads: plans back to back from four threads at once
    {
        std::vector<std::thread> th;
        std::atomic<int> bad{0};
        for (int t = 0; t < 4; ++t)
            th.emplace_back([&, t] {
                for (int r = 0; r < 6; ++r) {
                    const Csr &c = mats[(size_t)(t + r) % mats.size()];
                    dasp_options_t opt; dasp_options_default(&opt); opt.host_threads = 2 + t; opt.col_panels = (r & 1) ? 2 : 1;
                    dasp_plan_t *p = nullptr;
                    if (dasp_plan_create(&p, (r & 2) ? 16 : 64, c.m, c.n, c.nnz, c.rp.data(), c.ci.data(), (r & 2) ? (const void *)c.v16.data() : (const void *)c.v64.data(), &opt) != 0) ++bad;
                    dasp_plan_destroy(p);
                }
            });
        for (auto &x : th) x.join();
        CHECK(bad == 0);
    }
    // 4. the multi-GPU host part: every rank of a 3-way partition in the three overlap modes
    for (const char *name : {"HV15R", "Queen_4147"}) {
        const Csr c = synth(name, 0.01);
        int bounds[4];
        CHECK(dasp_partition_rows(c.m, c.rp.data(), 3, bounds) == 0);
        for (int mode = 0; mode <= 2; ++mode)
            for (int r = 0; r < 3; ++r) {
                const int r0 = bounds[r], r1 = bounds[r + 1];
                std::vector<int> rp((size_t)(r1 - r0) + 1);
                for (int i = r0; i <= r1; ++i) rp[(size_t)(i - r0)] = c.rp[(size_t)i] - c.rp[(size_t)r0];
                dasp_options_t opt; dasp_options_default(&opt); opt.cid16 = 1;
                dasp_mg_plan_t *mg = nullptr;
                CHECK(dasp_mg_plan_create(&mg, 64, c.m, c.n, 3, r, bounds, rp.data(), c.ci.data() + c.rp[(size_t)r0], c.v64.data() + c.rp[(size_t)r0], &opt, mode) == 0);
                dasp_mg_info_t info;
                if (mg) { CHECK(dasp_mg_info(mg, &info) == 0 && info.nnz_own + info.nnz_other == rp.back()); CHECK(dasp_mg_upload(mg) != 0); }
                dasp_mg_destroy(mg);
            }
    }
    std::printf("san_driver: %d files, %d matrices, %d failed checks\n", (int)files.size(), tag, fails);
    return fails ? 1 : 0;
}

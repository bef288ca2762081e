// mmio.cpp -- MatrixMarket -> CSR with the semantics of the reference's mmio_allinone
// (src/mmio_highlevel.h:608-774; banner and size-line rules of src/mmio.h:398-624):
//   * banner: five tokens, fields 2-5 case-folded; object "matrix"; format coordinate|array;
//     field real|complex|pattern|integer; symmetry general|symmetric|hermitian|skew-symmetric
//   * comment lines start with '%'; the size line is "M N nz" (blank lines before it tolerated)
//   * entries are whitespace-separated tokens (fscanf semantics): "i j v" (real), "i j re im"
//     (complex: real part kept), "i j iv" (integer, read as int), "i j" (pattern, value 1)
//   * symmetric OR hermitian input is mirrored (skew-symmetric is not), the mirrored entry
//     placed right after its source entry; rows keep file order; duplicates are kept
//   * return codes 0 / -1 (open) / -2 (banner) / -4 (size line)
// The whole file is read once and tokenised in memory instead of one fscanf per entry
// (the reference's loader is serial text parsing: minutes on 3e8-entry files).
#include <algorithm>
#include <cctype>
#include <cerrno>
#include <charconv>
#include <cstdio>
#include <cstdlib>
#include <chrono>
#include <cstring>
#include <memory>
#include <new>
#include <string>
#include <thread>
#include <vector>

#include "plan.hpp"

namespace dasp {

namespace {

// C-locale isspace without the libc call
static inline bool is_ws(char ch) { return ch == ' ' || (unsigned)(ch - '\t') <= 4u; }

struct Cursor {
    const char *p, *end;
    void skip_ws() { while (p < end && is_ws(*p)) ++p; }
    bool at_end() { skip_ws(); return p >= end; }
};

// "%d": optional sign, decimal digits; stops at the first non-digit
bool parse_int(Cursor &c, int &out)
{
    c.skip_ws();
    const char *q = c.p;
    bool neg = false;
    if (q < c.end && (*q == '+' || *q == '-')) { neg = *q == '-'; ++q; }
    if (q >= c.end || !std::isdigit((unsigned char)*q)) return false;
    long long v = 0;
    while (q < c.end && std::isdigit((unsigned char)*q)) { v = v * 10 + (*q - '0'); if (v > (1LL << 40)) v = 1LL << 40; ++q; }
    out = (int)(neg ? -v : v);
    c.p = q;
    return true;
}

// "%lg": strtod grammar (decimal, exponent, inf/nan, hex floats)
bool parse_double(Cursor &c, double &out)
{
    c.skip_ws();
    if (c.p >= c.end) return false;
    const char *q = c.p;
    if (*q == '+') ++q;   // from_chars rejects a leading '+', strtod accepts it
    auto r = std::from_chars(q, c.end, out, std::chars_format::general);
    if (r.ec == std::errc() && r.ptr != q) {
        // from_chars stops before things strtod would still accept (hex floats, "infinity" tails);
        // those only start with 0x / letters, which the plain path never consumes partially.
        if (!((r.ptr < c.end) && (*r.ptr == 'x' || *r.ptr == 'X'))) { c.p = r.ptr; return true; }
    }
    // slow path: the buffer is NUL-terminated by the caller
    char *endp = nullptr;
    errno = 0;
    double v = std::strtod(c.p, &endp);
    if (endp == c.p) return false;
    out = v;
    c.p = endp;
    return true;
}

// reads one line (like fgets with a 1025-byte buffer the reference uses: longer lines are split)
bool next_line(Cursor &c, std::string &line)
{
    if (c.p >= c.end) return false;
    const char *q = c.p;
    size_t n = 0;
    while (q < c.end && n < 1024) { ++n; if (*q++ == '\n') break; }
    line.assign(c.p, q);
    c.p = q;
    return true;
}

void lower(std::string &s) { for (auto &ch : s) ch = (char)std::tolower((unsigned char)ch); }

// COO (file order) -> CSR exactly as the reference's counting pass + scatter pass do it
// (mmio_highlevel.h:702-756), parallel over row ranges: every thread streams over all entries but only
// counts / places those whose row -- or mirrored row -- it owns, so the order inside a row stays file order.
template <class T>
int finish(int M, int N, int nz, bool sym, const raw_vector<int> &ri, const raw_vector<int> &cj,
           const raw_vector<double> &vv, int *m, int *n, int *nnz, int *symflag, int **rp_out, int **ci_out, void **val_out)
{
    int nthreads = nz < (1 << 20) ? 1 : resolve_threads(0);
    nthreads = std::min(nthreads, std::max(1, M));
    auto lo_of = [&](int t) { return (int)((long long)M * t / nthreads); };
    std::vector<int> cnt((size_t)M + 1, 0);
    auto run_par = [&](auto fn) {
        if (nthreads == 1) { fn(0); return; }
        std::vector<std::thread> th;
        for (int t = 0; t < nthreads; ++t) th.emplace_back(fn, t);
        for (auto &x : th) x.join();
    };
    run_par([&](int t) {
        const int lo = lo_of(t), hi = lo_of(t + 1);
        for (int e = 0; e < nz; ++e) {
            const int r = ri[e], c = cj[e];
            if (r >= lo && r < hi) cnt[r]++;
            if (sym && r != c && c >= lo && c < hi) cnt[c]++;
        }
    });
    int *rp = (int *)std::malloc(sizeof(int) * ((size_t)M + 1));
    if (!rp) return DASP_ERR_NOMEM;
    long long run = 0;
    for (int i = 0; i < M; ++i) { rp[i] = (int)run; run += cnt[i]; }
    if (run >= (1LL << 31)) { std::free(rp); set_error("nnz after mirroring exceeds int32"); return DASP_ERR_ENTRY; }
    rp[M] = (int)run;
    const size_t total = (size_t)run;
    int *ci = (int *)std::malloc(sizeof(int) * (total ? total : 1));
    T *v = (T *)std::malloc(sizeof(T) * (total ? total : 1));
    if (!ci || !v) { std::free(rp); std::free(ci); std::free(v); return DASP_ERR_NOMEM; }
    run_par([&](int t) {
        const int lo = lo_of(t), hi = lo_of(t + 1);
        for (int i = lo; i < hi; ++i) cnt[i] = 0;
        for (int e = 0; e < nz; ++e) {
            const int r = ri[e], c = cj[e];
            if (r >= lo && r < hi) { const size_t at = (size_t)rp[r] + cnt[r]++; ci[at] = c; v[at] = (T)vv[e]; }
            if (sym && r != c && c >= lo && c < hi) { const size_t at = (size_t)rp[c] + cnt[c]++; ci[at] = r; v[at] = (T)vv[e]; }
        }
    });
    *m = M; *n = N; *nnz = (int)total; *symflag = sym ? 1 : 0;
    *rp_out = rp; *ci_out = ci; *val_out = v;
    return DASP_OK;
}

}  // namespace

int load_mtx(const char *path, int precision, int *m, int *n, int *nnz, int *symflag, int **rp_out, int **ci_out, void **val_out)
{
    if (!path || !m || !n || !nnz || !symflag || !rp_out || !ci_out || !val_out) return DASP_ERR_ARG;
    const bool verbose = std::getenv("DASP_VERBOSE") != nullptr;
    auto tick = std::chrono::steady_clock::now();
    auto lap = [&](const char *what) { if (verbose) { auto now = std::chrono::steady_clock::now(); std::fprintf(stderr, "[dasp loader] %s: %.3f s\n", what, std::chrono::duration<double>(now - tick).count()); tick = now; } };
    FILE *f = std::fopen(path, "rb");
    if (!f) { set_error(std::string("cannot open ") + path); return DASP_ERR_OPEN; }
    std::unique_ptr<char[]> buf;
    size_t got = 0;
    {
        std::fseek(f, 0, SEEK_END);
        long sz = std::ftell(f);
        std::fseek(f, 0, SEEK_SET);
        if (sz < 0) sz = 0;
        buf.reset(new (std::nothrow) char[(size_t)sz + 1]);     // not zero-filled
        if (!buf) { std::fclose(f); set_error("out of host memory"); return DASP_ERR_NOMEM; }
        got = sz ? std::fread(buf.get(), 1, (size_t)sz, f) : 0;
        buf[got] = '\0';                                        // strtod slow path needs a terminator
        std::fclose(f);
    }
    Cursor c{buf.get(), buf.get() + got};
    lap("read file");

    // ---- banner (mmio.h:398-564)
    std::string line;
    if (!next_line(c, line)) { set_error("empty file"); return DASP_ERR_BANNER; }
    char t0[1025], t1[1025], t2[1025], t3[1025], t4[1025];
    if (std::sscanf(line.c_str(), "%1024s %1024s %1024s %1024s %1024s", t0, t1, t2, t3, t4) != 5) {
        set_error("banner: fewer than five fields"); return DASP_ERR_BANNER;
    }
    std::string obj(t1), fmt(t2), field(t3), symm(t4);
    lower(obj); lower(fmt); lower(field); lower(symm);
    if (std::strncmp(t0, "%%MatrixMarket", 14) != 0) { set_error("banner: missing %%MatrixMarket"); return DASP_ERR_BANNER; }
    if (obj != "matrix") { set_error("banner: object is not 'matrix'"); return DASP_ERR_BANNER; }
    if (fmt != "coordinate" && fmt != "array") { set_error("banner: unknown format"); return DASP_ERR_BANNER; }
    const bool is_real = field == "real", is_complex = field == "complex", is_pattern = field == "pattern",
               is_integer = field == "integer";
    if (!(is_real || is_complex || is_pattern || is_integer)) { set_error("banner: unknown field"); return DASP_ERR_BANNER; }
    if (symm != "general" && symm != "symmetric" && symm != "hermitian" && symm != "skew-symmetric") {
        set_error("banner: unknown symmetry"); return DASP_ERR_BANNER;
    }
    const bool sym = symm == "symmetric" || symm == "hermitian";   // mmio_highlevel.h:642

    // ---- size line (mmio.h:568-624)
    int M = 0, N = 0, nz = 0;
    do {
        if (!next_line(c, line)) { set_error("size line missing"); return DASP_ERR_SIZE; }
    } while (!line.empty() && line[0] == '%');
    if (std::sscanf(line.c_str(), "%d %d %d", &M, &N, &nz) != 3) {
        // blank (or short) line: the next three integer tokens of the stream are M N nz.
        // (the reference's fscanf loop never terminates on a non-numeric token; here it is an error)
        if (!(parse_int(c, M) && parse_int(c, N) && parse_int(c, nz))) {
            set_error("size line missing or malformed"); return DASP_ERR_SIZE;
        }
    }
    if (M < 0 || N < 0 || nz < 0) { set_error("negative dimension"); return DASP_ERR_SIZE; }

    // ---- entries (mmio_highlevel.h:663-697).  The entry section is a stream of whitespace-separated tokens
    // (fscanf semantics: an entry may span lines), tpe tokens per entry.  Parsed in parallel: the buffer is cut at
    // token boundaries, tokens are counted per piece, and with the prefix sums every piece knows which
    // (entry, field) each of its tokens is.
    const int tpe = is_pattern ? 2 : (is_complex ? 4 : 3);
    const long long need = (long long)nz * tpe;
    c.skip_ws();
    const char *const b0 = c.p, *const b1 = c.end;
    int nthreads = resolve_threads(0);
    if (b1 - b0 < (1 << 20)) nthreads = 1;
    std::vector<const char *> cut((size_t)nthreads + 1);
    cut[0] = b0; cut[nthreads] = b1;
    for (int t = 1; t < nthreads; ++t) {
        const char *q = b0 + (b1 - b0) * t / nthreads;
        while (q < b1 && !is_ws(*q)) ++q;   // finish the token we landed in
        cut[t] = q;
    }
    for (int t = 1; t <= nthreads; ++t) if (cut[t] < cut[t - 1]) cut[t] = cut[t - 1];
    std::vector<long long> ntok((size_t)nthreads + 1, 0);
    auto count_tokens = [&](int t) {
        const char *q = cut[t], *e = cut[t + 1];
        long long k = 0;
        while (q < e) {
            while (q < e && is_ws(*q)) ++q;
            if (q >= e) break;
            ++k;
            while (q < e && !is_ws(*q)) ++q;
        }
        ntok[t + 1] = k;
    };
    {
        std::vector<std::thread> th;
        for (int t = 0; t < nthreads; ++t) th.emplace_back(count_tokens, t);
        for (auto &x : th) x.join();
    }
    for (int t = 0; t < nthreads; ++t) ntok[t + 1] += ntok[t];
    lap("count tokens");
    if (ntok[nthreads] < need) {
        set_error("entry " + std::to_string(ntok[nthreads] / tpe + 1) + ": malformed or missing"); return DASP_ERR_ENTRY;
    }
    // only now (the file is known to hold nz entries, so nz <= bytes / 4) are the nz-sized arrays allocated: a size line
    // promising 2 G entries in a 30-byte file never reaches an allocation.  Not zero-filled: every slot is parsed into.
    raw_vector<int> ri, cj;
    raw_vector<double> vv;
    try { ri.resize((size_t)nz); cj.resize((size_t)nz); vv.resize((size_t)nz); }
    catch (const std::bad_alloc &) { set_error("out of host memory"); return DASP_ERR_NOMEM; }
    std::vector<long long> bad_entry((size_t)nthreads, -1);
    std::vector<int> bad_kind((size_t)nthreads, 0);
    auto parse_piece = [&](int t) {
        Cursor cc{cut[t], cut[t + 1]};
        long long g = ntok[t];
        while (g < need) {
            cc.skip_ws();
            if (cc.p >= cc.end) break;
            const long long e = g / tpe;
            const int field = (int)(g % tpe);
            bool ok = true;
            const char *tok = cc.p;
            if (field == 0) { int i; ok = parse_int(cc, i); if (ok) { if (i < 1 || i > M) { bad_entry[t] = e; bad_kind[t] = 2; return; } ri[e] = i - 1; } }
            else if (field == 1) {
                int j; ok = parse_int(cc, j);
                if (ok) { if (j < 1 || j > N || (sym && j > M)) { bad_entry[t] = e; bad_kind[t] = 2; return; } cj[e] = j - 1; }
                if (ok && is_pattern) vv[e] = 1.0;
            } else if (field == 2) {
                if (is_integer) { int iv; ok = parse_int(cc, iv); if (ok) vv[e] = iv; }
                else { double re; ok = parse_double(cc, re); if (ok) vv[e] = re; }
            } else { double im; ok = parse_double(cc, im); }
            // a token must be consumed entirely (fscanf would leave the rest for the next conversion and derail)
            if (ok && cc.p < cc.end && !is_ws(*cc.p)) ok = false;
            if (!ok) { bad_entry[t] = e; bad_kind[t] = 1; (void)tok; return; }
            ++g;
        }
    };
    {
        std::vector<std::thread> th;
        for (int t = 0; t < nthreads; ++t) th.emplace_back(parse_piece, t);
        for (auto &x : th) x.join();
    }
    for (int t = 0; t < nthreads; ++t)
        if (bad_entry[t] >= 0) {
            set_error("entry " + std::to_string(bad_entry[t] + 1) + (bad_kind[t] == 2 ? ": index out of range" : ": malformed or missing"));
            return DASP_ERR_ENTRY;
        }
    lap("parse tokens");
    const int rc = precision == 64 ? finish<double>(M, N, nz, sym, ri, cj, vv, m, n, nnz, symflag, rp_out, ci_out, val_out)
                                   : finish<_Float16>(M, N, nz, sym, ri, cj, vv, m, n, nnz, symflag, rp_out, ci_out, val_out);
    lap("coo -> csr");
    return rc;
}

// ---- binary CSR cache: the CSR exactly as the loader returns it, so a second run skips text parsing.
// layout: "DASPCSR1" | int32 precision, m, n, nnz, is_symmetric, 0,0,0 | rowptr[m+1] | colidx[nnz] | val[nnz]
static const char kMagic[8] = {'D', 'A', 'S', 'P', 'C', 'S', 'R', '1'};

int save_csr_bin(const char *path, int precision, int m, int n, int nnz, int sym, const int *rp, const int *ci, const void *val)
{
    if (!path || !rp || (nnz > 0 && (!ci || !val)) || (precision != 64 && precision != 16)) return DASP_ERR_ARG;
    FILE *f = std::fopen(path, "wb");
    if (!f) { set_error(std::string("cannot create ") + path); return DASP_ERR_OPEN; }
    const int hdr[8] = {precision, m, n, nnz, sym, 0, 0, 0};
    const size_t vb = precision == 64 ? 8 : 2;
    bool ok = std::fwrite(kMagic, 1, 8, f) == 8 && std::fwrite(hdr, 4, 8, f) == 8 &&
              std::fwrite(rp, 4, (size_t)m + 1, f) == (size_t)m + 1 &&
              (nnz == 0 || (std::fwrite(ci, 4, (size_t)nnz, f) == (size_t)nnz && std::fwrite(val, vb, (size_t)nnz, f) == (size_t)nnz));
    ok = (std::fclose(f) == 0) && ok;
    if (!ok) { set_error(std::string("short write to ") + path); return DASP_ERR_OPEN; }
    return DASP_OK;
}

int load_csr_bin(const char *path, int precision, int *m, int *n, int *nnz, int *sym, int **rp_out, int **ci_out, void **val_out)
{
    if (!path || !m || !n || !nnz || !sym || !rp_out || !ci_out || !val_out) return DASP_ERR_ARG;
    FILE *f = std::fopen(path, "rb");
    if (!f) { set_error(std::string("cannot open ") + path); return DASP_ERR_OPEN; }
    char magic[8]; int hdr[8];
    if (std::fread(magic, 1, 8, f) != 8 || std::memcmp(magic, kMagic, 8) != 0 || std::fread(hdr, 4, 8, f) != 8 || hdr[0] != precision ||
        hdr[1] < 0 || hdr[2] < 0 || hdr[3] < 0) {
        std::fclose(f); set_error("not a DASPCSR1 file of this precision"); return DASP_ERR_BANNER;
    }
    const size_t M = (size_t)hdr[1], K = (size_t)hdr[3], vb = precision == 64 ? 8 : 2;
    int *rp = (int *)std::malloc(4 * (M + 1)), *ci = (int *)std::malloc(4 * (K ? K : 1));
    void *v = std::malloc(vb * (K ? K : 1));
    bool ok = rp && ci && v && std::fread(rp, 4, M + 1, f) == M + 1 && (K == 0 || (std::fread(ci, 4, K, f) == K && std::fread(v, vb, K, f) == K));
    std::fclose(f);
    if (!ok || rp[0] != 0 || rp[M] != hdr[3]) { std::free(rp); std::free(ci); std::free(v); set_error("truncated or corrupt DASPCSR1 file"); return DASP_ERR_ENTRY; }
    *m = hdr[1]; *n = hdr[2]; *nnz = hdr[3]; *sym = hdr[4];
    *rp_out = rp; *ci_out = ci; *val_out = v;
    return DASP_OK;
}

}  // namespace dasp

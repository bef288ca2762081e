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
#include <cctype>
#include <cerrno>
#include <charconv>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "plan.hpp"

namespace dasp {

namespace {

struct Cursor {
    const char *p, *end;
    void skip_ws() { while (p < end && std::isspace((unsigned char)*p)) ++p; }
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

template <class T>
int finish(int M, int N, int nz, bool sym, const std::vector<int> &ri, const std::vector<int> &cj,
           const std::vector<double> &vv, int *m, int *n, int *nnz, int *symflag, int **rp_out, int **ci_out, void **val_out)
{
    std::vector<int> cnt((size_t)M + 1, 0);
    for (int e = 0; e < nz; ++e) cnt[ri[e]]++;
    if (sym) for (int e = 0; e < nz; ++e) if (ri[e] != cj[e]) cnt[cj[e]]++;
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
    std::fill(cnt.begin(), cnt.end(), 0);
    for (int e = 0; e < nz; ++e) {
        const int r = ri[e], c = cj[e];
        size_t at = (size_t)rp[r] + cnt[r]++;
        ci[at] = c; v[at] = (T)vv[e];
        if (sym && r != c) { at = (size_t)rp[c] + cnt[c]++; ci[at] = r; v[at] = (T)vv[e]; }
    }
    *m = M; *n = N; *nnz = (int)total; *symflag = sym ? 1 : 0;
    *rp_out = rp; *ci_out = ci; *val_out = v;
    return DASP_OK;
}

}  // namespace

int load_mtx(const char *path, int precision, int *m, int *n, int *nnz, int *symflag, int **rp_out, int **ci_out, void **val_out)
{
    if (!path || !m || !n || !nnz || !symflag || !rp_out || !ci_out || !val_out) return DASP_ERR_ARG;
    FILE *f = std::fopen(path, "rb");
    if (!f) { set_error(std::string("cannot open ") + path); return DASP_ERR_OPEN; }
    std::vector<char> buf;
    {
        std::fseek(f, 0, SEEK_END);
        long sz = std::ftell(f);
        std::fseek(f, 0, SEEK_SET);
        if (sz < 0) sz = 0;
        buf.resize((size_t)sz + 1);
        size_t got = sz ? std::fread(buf.data(), 1, (size_t)sz, f) : 0;
        buf.resize(got + 1);
        buf[got] = '\0';
        std::fclose(f);
    }
    Cursor c{buf.data(), buf.data() + buf.size() - 1};

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

    // ---- entries (mmio_highlevel.h:663-697)
    std::vector<int> ri((size_t)nz), cj((size_t)nz);
    std::vector<double> vv((size_t)nz);
    for (int e = 0; e < nz; ++e) {
        int i, j, iv = 0;
        double re = 1.0, im = 0.0;
        bool ok = parse_int(c, i) && parse_int(c, j);
        if (ok && is_real) ok = parse_double(c, re);
        else if (ok && is_complex) ok = parse_double(c, re) && parse_double(c, im);
        else if (ok && is_integer) { ok = parse_int(c, iv); re = iv; }
        if (!ok) { set_error("entry " + std::to_string(e + 1) + ": malformed or missing"); return DASP_ERR_ENTRY; }
        --i; --j;
        if (i < 0 || i >= M || j < 0 || j >= N || (sym && j >= M) ) {
            set_error("entry " + std::to_string(e + 1) + ": index out of range"); return DASP_ERR_ENTRY;
        }
        ri[e] = i; cj[e] = j; vv[e] = re;
    }
    if (precision == 64)
        return finish<double>(M, N, nz, sym, ri, cj, vv, m, n, nnz, symflag, rp_out, ci_out, val_out);
    return finish<_Float16>(M, N, nz, sym, ri, cj, vv, m, n, nnz, symflag, rp_out, ci_out, val_out);
}

}  // namespace dasp

// planio.cpp -- serialised plans (the reference never stores its packed format).
// A plan file holds every host array of a Plan, so a later run (or every rank of a multi-GPU job)
// skips classification and packing: load -> upload -> spmv.
// layout: "DASPPLN3" | plan, where plan = int32 precision, m, n, nnz, y_order, windowed, row_window, lds_bytes, cid16, n_parts,
//         part_stride, stream_policy, n_panels, n_mfma_rows | dasp_stats_t | ShortGroup[5] | for each array, in a fixed order: int64 byte
//         count + bytes | the n_panels column panels, each a nested plan
#include <cstdio>
#include <cstring>

#include "plan.hpp"

namespace dasp {

namespace {
const char kPlanMagic[8] = {'D', 'A', 'S', 'P', 'P', 'L', 'N', '3'};

struct Writer {
    FILE *f; bool ok = true;
    void raw(const void *p, size_t n) { if (ok && n && std::fwrite(p, 1, n, f) != n) ok = false; }
    template <class V> void vec(const V &v) { long long b = (long long)(v.size() * sizeof(typename V::value_type)); raw(&b, 8); raw(v.data(), (size_t)b); }
};
struct Reader {
    FILE *f; bool ok = true;
    void raw(void *p, size_t n) { if (ok && n && std::fread(p, 1, n, f) != n) ok = false; }
    template <class V> void vec(V &v)
    {
        long long b = -1; raw(&b, 8);
        if (!ok || b < 0 || b % (long long)sizeof(typename V::value_type)) { ok = false; return; }
        v.resize((size_t)b / sizeof(typename V::value_type));
        raw(v.data(), (size_t)b);
    }
};

template <class IO> void arrays(IO &io, Plan &p)
{
    io.vec(p.part_bounds); io.vec(p.order); io.vec(p.dst_map); io.vec(p.panel_bounds);
    io.vec(p.long_val); io.vec(p.long_cid); io.vec(p.piece_ptr); io.vec(p.piece_dst); io.vec(p.multi_ptr); io.vec(p.multi_dst);
    io.vec(p.med_ptr); io.vec(p.med_val); io.vec(p.med_cid); io.vec(p.med_cid16); io.vec(p.med_base);
    io.vec(p.irr_ptr); io.vec(p.irr_val); io.vec(p.irr_cid);
    io.vec(p.med_dst); io.vec(p.win_cmin); io.vec(p.win_len);
    io.vec(p.short_val); io.vec(p.short_cid);
}
}  // namespace

static void write_plan(Writer &w, Plan &p)
{
    const int hdr[14] = {p.precision, p.m, p.n, p.nnz, p.opt.y_order, p.windowed ? 1 : 0, p.row_window, p.lds_bytes, p.cid16 ? 1 : 0,
                         p.opt.n_parts, p.opt.part_stride, p.opt.stream_policy, (int)p.panels.size(), p.n_mfma_rows};
    w.raw(hdr, sizeof hdr); w.raw(&p.stats, sizeof p.stats); w.raw(p.grp, sizeof p.grp);
    arrays(w, p);
    for (auto &h : p.panels) write_plan(w, h->impl);
}

static bool read_plan(Reader &r, Plan &p, int depth)
{
    int hdr[14];
    r.raw(hdr, sizeof hdr);
    if (!r.ok || (hdr[0] != 64 && hdr[0] != 16) || hdr[12] < 0 || hdr[12] > 64 || (depth > 0 && hdr[12] != 0)) return false;
    p.precision = hdr[0]; p.geo = geometry_for(p.precision);
    p.m = hdr[1]; p.n = hdr[2]; p.nnz = hdr[3];
    dasp_options_default(&p.opt);
    p.opt.y_order = hdr[4]; p.windowed = hdr[5] != 0; p.row_window = hdr[6]; p.lds_bytes = hdr[7]; p.cid16 = hdr[8] != 0;
    p.opt.n_parts = hdr[9]; p.opt.part_stride = hdr[10]; p.opt.stream_policy = hdr[11]; p.n_mfma_rows = hdr[13];
    r.raw(&p.stats, sizeof p.stats); r.raw(p.grp, sizeof p.grp);
    arrays(r, p);
    if (!r.ok) return false;
    p.opt.part_bounds = p.part_bounds.empty() ? nullptr : p.part_bounds.data();
    const size_t vb = (size_t)p.geo.vbytes;
    const int np = hdr[12];
    bool sane = p.m >= 0 && p.order.size() == (size_t)p.m && p.stats.rowA == p.m && p.stats.precision == p.precision &&
                (p.dst_map.empty() || p.dst_map.size() == (size_t)p.m) && p.panel_bounds.size() == 2 * (size_t)np &&
                p.long_val.size() == p.long_cid.size() * vb && p.irr_val.size() == p.irr_cid.size() * vb &&
                p.short_val.size() == p.short_cid.size() * vb &&
                (p.cid16 ? p.med_val.size() == p.med_cid16.size() * vb : p.med_val.size() == p.med_cid.size() * vb);
    if (np == 0)   // a packed plan (a panel parent keeps none of the row-structure arrays)
        sane = sane && p.piece_ptr.size() == p.piece_dst.size() + 1 && p.irr_ptr.size() == (size_t)p.n_mfma_rows + 1 &&
               p.n_mfma_rows >= 0 && p.n_mfma_rows <= p.stats.row_block && (!p.windowed || p.med_dst.size() == (size_t)p.n_mfma_rows);
    if (!sane) return false;
    p.cnt_long = p.long_cid.size(); p.cnt_irr = p.irr_cid.size(); p.cnt_short = p.short_cid.size();
    p.cnt_reg = p.cid16 ? p.med_cid16.size() : p.med_cid.size();
    p.host_dropped = false;
    p.panel = depth > 0;
    p.opt.col_panels = np > 0 ? np : 1;
    for (int k = 0; k < np; ++k) {
        std::unique_ptr<dasp_plan> h(new dasp_plan());
        if (!read_plan(r, h->impl, depth + 1) || h->impl.m != p.m || h->impl.precision != p.precision) return false;
        p.panels.push_back(std::move(h));
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
    Reader r{f};
    char magic[8];
    r.raw(magic, 8);
    if (!r.ok || std::memcmp(magic, kPlanMagic, 8) != 0) { std::fclose(f); set_error("not a DASPPLN3 plan file"); return DASP_ERR_BANNER; }
    const bool ok = read_plan(r, p, 0);
    std::fclose(f);
    if (!ok) { set_error("truncated or inconsistent plan file"); return DASP_ERR_ENTRY; }
    return DASP_OK;
}

}  // namespace dasp

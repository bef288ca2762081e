// capi.cpp -- extern "C" surface of libdasp_amd.so (include/dasp_amd.h).
#include <hip/hip_runtime_api.h>

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <exception>
#include <memory>
#include <new>
#include <string>
#include <sys/stat.h>
#include <vector>

#include "plan.hpp"

namespace dasp {
const char *last_error_cstr();
int upload_plan(Plan &p);
int tune_placement(Plan &p, int trials, const void *dX, void *dY, double *ms_first, double *ms_kept);
int launch_spmv(Plan &p, const void *dX, void *dY, void *stream, bool accumulate);
int time_spmv_each(Plan &p, const void *dX, void *dY, void *stream, int warmup, int iters, float *ms_each);
int time_spmv(Plan &p, const void *dX, void *dY, void *stream, int warmup, int iters, double *wall_ms, double *event_ms);
int time_spmv_graph(Plan &p, const void *dX, void *dY, void *stream, int warmup, int iters, int batch, double *wall_ms, double *event_ms);
int selftest_mfma();
int set_stream_policy(Plan &p, int policy);
int download_array(Plan &p, const char *name, void *dst, size_t bytes);
}  // namespace dasp

using namespace dasp;

// no exception may cross the C ABI (ctypes / cgo / JNI callers would reach std::terminate): every entry point that can
// allocate runs inside this guard and reports DASP_ERR_NOMEM / DASP_ERR_ENTRY instead
template <class F>
static int guarded(const char *what, F f) noexcept
{
    try { return f(); }
    catch (const std::bad_alloc &) { set_error(std::string(what) + ": out of host memory"); return DASP_ERR_NOMEM; }
    catch (const std::exception &e) { set_error(std::string(what) + ": " + e.what()); return DASP_ERR_ENTRY; }
    catch (...) { set_error(std::string(what) + ": unknown exception"); return DASP_ERR_ENTRY; }
}

extern "C" {

const char *dasp_last_error(void) { return last_error_cstr(); }
const char *dasp_version(void) { return "dasp_amd 0.1 gfx950"; }
void dasp_free(void *p) { std::free(p); }

void dasp_options_default(dasp_options_t *o)
{
    if (!o) return;
    std::memset(o, 0, sizeof *o);
    o->threshold = 0.75;      // src/main_f64.cu:125
    o->block_longest = 256;   // src/main_f64.cu:124
    o->y_order = DASP_Y_PERMUTED;
}

int dasp_mmio_allinone_f64(int *m, int *n, int *nnz, int *isSymmetric, int **csrRowPtr, int **csrColIdx,
                           double **csrVal, const char *filename)
{
    void *v = nullptr;
    int rc = guarded("dasp_mmio_allinone_f64", [&] { return load_mtx(filename, 64, m, n, nnz, isSymmetric, csrRowPtr, csrColIdx, &v); });
    if (rc == DASP_OK) *csrVal = static_cast<double *>(v);
    return rc;
}

int dasp_mmio_allinone_f16(int *m, int *n, int *nnz, int *isSymmetric, int **csrRowPtr, int **csrColIdx,
                           uint16_t **csrVal, const char *filename)
{
    void *v = nullptr;
    int rc = guarded("dasp_mmio_allinone_f16", [&] { return load_mtx(filename, 16, m, n, nnz, isSymmetric, csrRowPtr, csrColIdx, &v); });
    if (rc == DASP_OK) *csrVal = static_cast<uint16_t *>(v);
    return rc;
}

int dasp_csr_save(const char *path, int precision, int m, int n, int nnz, int isSymmetric, const int *csrRowPtr,
                  const int *csrColIdx, const void *csrVal)
{
    return guarded("dasp_csr_save", [&] { return save_csr_bin(path, precision, m, n, nnz, isSymmetric, csrRowPtr, csrColIdx, csrVal); });
}

int dasp_csr_load(const char *path, int precision, int *m, int *n, int *nnz, int *isSymmetric, int **csrRowPtr,
                  int **csrColIdx, void **csrVal)
{
    return guarded("dasp_csr_load", [&] { return load_csr_bin(path, precision, m, n, nnz, isSymmetric, csrRowPtr, csrColIdx, csrVal); });
}

// options as the builders expect them: defaults for unset fields, a private, validated copy of the column partition
static int normalise_options(Plan &p, const dasp_options_t *opt, int colA)
{
    if (opt) p.opt = *opt; else dasp_options_default(&p.opt);
    if (!(p.opt.threshold > 0)) p.opt.threshold = 0.75;
    if (p.opt.block_longest < 6) p.opt.block_longest = 256;
    if (p.opt.n_parts > 0) {
        if (!p.opt.part_bounds || p.opt.part_stride <= 0) { set_error("bad column partition"); return DASP_ERR_ARG; }
        p.part_bounds.assign(p.opt.part_bounds, p.opt.part_bounds + p.opt.n_parts + 1);
        if (p.part_bounds.front() != 0 || p.part_bounds.back() != colA) { set_error("part_bounds must span [0,colA]"); return DASP_ERR_ARG; }
        for (int g = 0; g < p.opt.n_parts; ++g)
            if (p.part_bounds[g + 1] < p.part_bounds[g] || p.part_bounds[g + 1] - p.part_bounds[g] > p.opt.part_stride) {
                set_error("part_bounds not monotone or wider than part_stride"); return DASP_ERR_ARG;
            }
        p.opt.part_bounds = p.part_bounds.data();
    } else { p.opt.n_parts = 0; p.opt.part_bounds = nullptr; }
    return DASP_OK;
}

int dasp_plan_create(dasp_plan_t **out, int precision, int rowA, int colA, int nnzA, const int *rp, const int *ci,
                     const void *val, const dasp_options_t *opt)
{
    if (!out) return DASP_ERR_ARG;
    *out = nullptr;
    if ((precision != 64 && precision != 16) || rowA < 0 || colA < 0 || nnzA < 0 || !rp || (nnzA > 0 && (!ci || !val))) {
        set_error("dasp_plan_create: bad arguments");
        return DASP_ERR_ARG;
    }
    return guarded("dasp_plan_create", [&] {
        std::unique_ptr<dasp_plan> h(new dasp_plan());
        Plan &p = h->impl;
        p.precision = precision; p.m = rowA; p.n = colA; p.nnz = nnzA;
        if (int rc = normalise_options(p, opt, colA)) return rc;
        if (int rc = build_plan(p, rp, ci, val)) return rc;
        *out = h.release();
        return (int)DASP_OK;
    });
}

// CSR already on the GPU: row pointer to the host (4(m+1) bytes), every O(rows) decision there, the nonzeros never leave the device
int dasp_plan_create_device(dasp_plan_t **out, int precision, int rowA, int colA, int nnzA, const int *dRowPtr, const int *dColIdx,
                            const void *dVal, const dasp_options_t *opt)
{
    if (!out) return DASP_ERR_ARG;
    *out = nullptr;
    if ((precision != 64 && precision != 16) || rowA < 0 || colA < 0 || nnzA < 0 || !dRowPtr || (nnzA > 0 && (!dColIdx || !dVal))) {
        set_error("dasp_plan_create_device: bad arguments");
        return DASP_ERR_ARG;
    }
    return guarded("dasp_plan_create_device", [&] {
        std::vector<int> rp((size_t)rowA + 1);
        if (hipMemcpy(rp.data(), dRowPtr, sizeof(int) * ((size_t)rowA + 1), hipMemcpyDeviceToHost) != hipSuccess) {
            set_error("cannot read the device row pointer (no HIP device, or not a device pointer)");
            return (int)DASP_ERR_HIP;
        }
        std::unique_ptr<dasp_plan> h(new dasp_plan());
        Plan &p = h->impl;
        p.precision = precision; p.m = rowA; p.n = colA; p.nnz = nnzA;
        if (int rc = normalise_options(p, opt, colA)) return rc;
        const DevCsr dev{dRowPtr, dColIdx, dVal};
        if (int rc = build_plan(p, rp.data(), nullptr, nullptr, &dev)) return rc;
        *out = h.release();
        return (int)DASP_OK;
    });
}

int dasp_plan_download_array(dasp_plan_t *plan, const char *name, void *dst, size_t bytes)
{
    if (!plan || !name || (!dst && bytes)) return DASP_ERR_ARG;
    return guarded("dasp_plan_download_array", [&] { return download_array(plan->impl, name, dst, bytes); });
}

void dasp_plan_destroy(dasp_plan_t *plan) { delete plan; }

int dasp_plan_save(dasp_plan_t *plan, const char *path)
{
    if (!plan) return DASP_ERR_ARG;
    return guarded("dasp_plan_save", [&] { return save_plan(plan->impl, path); });
}

int dasp_plan_load(dasp_plan_t **out, const char *path)
{
    if (!out) return DASP_ERR_ARG;
    *out = nullptr;
    return guarded("dasp_plan_load", [&] {
        std::unique_ptr<dasp_plan> h(new dasp_plan());
        if (int rc = load_plan(h->impl, path)) return rc;
        *out = h.release();
        return (int)DASP_OK;
    });
}

const int *dasp_plan_order(const dasp_plan_t *plan) { return plan ? plan->impl.order.data() : nullptr; }

int dasp_plan_y_order(const dasp_plan_t *plan) { return plan ? plan->impl.opt.y_order : DASP_ERR_ARG; }

long long dasp_plan_x_len(const dasp_plan_t *plan)
{
    if (!plan) return DASP_ERR_ARG;
    const Plan &p = plan->impl;
    return p.opt.n_parts > 0 ? (long long)p.opt.n_parts * p.opt.part_stride : (long long)p.n;
}

int dasp_plan_stats(const dasp_plan_t *plan, dasp_stats_t *out)
{
    if (!plan || !out) return DASP_ERR_ARG;
    *out = plan->impl.stats;
    return DASP_OK;
}

int dasp_plan_panel_count(const dasp_plan_t *plan) { return plan ? (int)plan->impl.panels.size() : DASP_ERR_ARG; }

dasp_plan_t *dasp_plan_panel(dasp_plan_t *plan, int k)
{
    if (!plan || k < 0 || k >= (int)plan->impl.panels.size()) { set_error("no such column panel"); return nullptr; }
    return plan->impl.panels[(size_t)k].get();
}

int dasp_plan_panel_range(const dasp_plan_t *plan, int k, int *col_begin, int *col_end)
{
    if (!plan || k < 0 || k >= (int)plan->impl.panels.size() || !col_begin || !col_end) { set_error("no such column panel"); return DASP_ERR_ARG; }
    *col_begin = plan->impl.panel_bounds[2 * (size_t)k]; *col_end = plan->impl.panel_bounds[2 * (size_t)k + 1];
    return DASP_OK;
}

static long long host_array_impl(const dasp_plan_t *plan, const char *name, const void **ptr, int *elem_bytes);
long long dasp_plan_host_array(const dasp_plan_t *plan, const char *name, const void **ptr, int *elem_bytes)
{
    if (!plan || !name || !ptr || !elem_bytes) return DASP_ERR_ARG;
    try { return host_array_impl(plan, name, ptr, elem_bytes); }
    catch (const std::exception &e) { set_error(std::string("dasp_plan_host_array: ") + e.what()); return DASP_ERR_NOMEM; }
}
static long long host_array_impl(const dasp_plan_t *plan, const char *name, const void **ptr, int *elem_bytes)
{
    const Plan &p = plan->impl;
    // the nnz-sized arrays exist on the host only until dasp_plan_drop_host (never, for a plan packed on the device);
    // the O(rows) arrays and order_rid always do
    static const char *const kBulk[] = {"long_val", "long_cid", "long_cid16", "med_val", "med_cid", "med_cid16", "med_cid8", "irr_val", "irr_cid", "short_val", "short_cid", "rt_val", "rt_cid",
                                        "tp_val", "tp_lrow", "tp_lcol", "tp_dst", "lcb_val", "lcb_lcol"};
    if (p.host_dropped)
        for (const char *b : kBulk)
            if (std::strcmp(name, b) == 0) { set_error("host copy of this array was dropped (use dasp_plan_download_array)"); return DASP_ERR_STATE; }
    const int vb = p.geo.vbytes;
    auto ints = [&](const std::vector<int> &v) { *ptr = v.data(); *elem_bytes = 4; return (long long)v.size(); };
    auto vals = [&](const raw_vector<char> &v) { *ptr = v.data(); *elem_bytes = vb; return (long long)(v.size() / vb); };
    auto rints = [&](const raw_vector<int> &v) { *ptr = v.data(); *elem_bytes = 4; return (long long)v.size(); };
    const std::string n(name);
    if (n == "order") return ints(p.order);
    if (n == "dst_map") return ints(p.dst_map);
    if (n == "long_val") return vals(p.long_val);
    if (n == "long_cid") return rints(p.long_cid);
    if (n == "long_cid16") { *ptr = p.long_cid16.data(); *elem_bytes = 2; return (long long)p.long_cid16.size(); }
    if (n == "long_base") {
        if (p.long_base.empty() && p.cnt_long_chunks > 0) { set_error("long_base lives on the device only (use dasp_plan_download_array)"); return DASP_ERR_STATE; }
        return ints(p.long_base);
    }
    if (n == "piece_c16") return ints(p.piece_c16);
    if (n == "piece_ptr") return ints(p.piece_ptr);
    if (n == "piece_dst") return ints(p.piece_dst);
    if (n == "multi_ptr") return ints(p.multi_ptr);
    if (n == "multi_dst") return ints(p.multi_dst);
    if (n == "med_ptr") return ints(p.med_ptr);
    if (n == "med_val") return vals(p.med_val);
    if (n == "med_cid") return rints(p.med_cid);
    if (n == "irr_ptr") return ints(p.irr_ptr);
    if (n == "irr_val") return vals(p.irr_val);
    if (n == "irr_cid") return rints(p.irr_cid);
    if (n == "med_cid16") { *ptr = p.med_cid16.data(); *elem_bytes = 2; return (long long)p.med_cid16.size(); }
    if (n == "med_cid8") { *ptr = p.med_cid8.data(); *elem_bytes = 1; return (long long)p.med_cid8.size(); }
    if (n == "med_c8ptr") return ints(p.med_c8ptr);
    if (n == "med_korig") return ints(p.med_korig);
    if (n == "med_base") {
        if (p.cid16 && p.med_base.empty() && p.med_ptr.back() > 0) { set_error("med_base lives on the device only (use dasp_plan_download_array)"); return DASP_ERR_STATE; }
        return ints(p.med_base);
    }
    if (n == "med_dst") return ints(p.med_dst);
    if (n == "win_cmin") return ints(p.win_cmin);
    if (n == "win_len") return ints(p.win_len);
    if (n == "short_val") return vals(p.short_val);
    if (n == "short_cid") return rints(p.short_cid);
    if (n == "rt_val") return vals(p.rt_val);
    if (n == "rt_cid") return rints(p.rt_cid);
    if (n == "rt_ptr") return ints(p.rt_ptr);
    if (n == "rt_start") { *ptr = p.rt_start.data(); *elem_bytes = 2; return (long long)p.rt_start.size(); }
    if (n == "rt_mask") { *ptr = p.rt_mask.data(); *elem_bytes = 8; return (long long)p.rt_mask.size(); }
    if (n == "lcb_row_dst") return ints(p.lcb.row_dst);
    if (n == "lcb_row_id") return ints(p.lcb.row_id);
    if (n == "lcb_ptr") return ints(p.lcb.ptr);
    if (n == "lcb_unit") return ints(p.lcb.unit);
    if (n == "lcb_val") return vals(p.lcb.val);
    if (n == "lcb_lcol") { *ptr = p.lcb.lcol.data(); *elem_bytes = 2; return (long long)p.lcb.lcol.size(); }
    if (n == "tp_rb_row0") return ints(p.tp.rb_row0);
    if (n == "tp_rb_seg0") return ints(p.tp.rb_seg0);
    if (n == "tp_unit") return ints(p.tp.unit);
    if (n == "tp_dst") return ints(p.tp.dst);
    if (n == "tp_val") return vals(p.tp.val);
    if (n == "tp_lrow") { *ptr = p.tp.lrow.data(); *elem_bytes = 2; return (long long)p.tp.lrow.size(); }
    if (n == "tp_lcol") { *ptr = p.tp.lcol.data(); *elem_bytes = 2; return (long long)p.tp.lcol.size(); }
    if (n == "short_groups") {   // kNumShortGroups x {len,count,tiles,tile0,elem_off_lo,elem_off_hi,split,base0,base1,grp0,grp1,off0,off1,seg,rpt}
        static thread_local std::vector<int> flat;
        flat.clear();
        for (int g = 0; g < kNumShortGroups; ++g) {
            const ShortGroup &G = p.grp[g];
            const int row[15] = {G.len, G.count, G.tiles, G.tile0, (int)(G.elem_off & 0xffffffffll), (int)(G.elem_off >> 32),
                                 G.map.split, G.map.base[0], G.map.base[1], G.map.grp[0], G.map.grp[1], G.map.off[0], G.map.off[1], G.seg, G.rpt};
            flat.insert(flat.end(), row, row + 15);
        }
        return ints(flat);
    }
    set_error("unknown host array: " + n);
    return DASP_ERR_ARG;
}

int dasp_plan_upload(dasp_plan_t *plan)
{
    if (!plan) return DASP_ERR_ARG;
    return guarded("dasp_plan_upload", [&] { return upload_plan(plan->impl); });
}

int dasp_plan_tune_placement(dasp_plan_t *plan, int trials, const void *dX, void *dY, double *ms_first, double *ms_kept)
{
    if (!plan) return DASP_ERR_ARG;
    if (!plan->impl.dev) { set_error("plan not uploaded"); return DASP_ERR_STATE; }
    return guarded("dasp_plan_tune_placement", [&] { return tune_placement(plan->impl, trials, dX, dY, ms_first, ms_kept); });
}

int dasp_plan_drop_host(dasp_plan_t *plan)
{
    if (!plan) return DASP_ERR_ARG;
    Plan &p = plan->impl;
    if (!p.dev) { set_error("upload the plan before dropping its host arrays"); return DASP_ERR_STATE; }
    auto dropc = [](raw_vector<char> &v) { raw_vector<char>().swap(v); };
    auto dropi = [](raw_vector<int> &v) { raw_vector<int>().swap(v); };
    dropc(p.long_val); dropi(p.long_cid); raw_vector<uint16_t>().swap(p.long_cid16); dropc(p.med_val); dropi(p.med_cid); raw_vector<uint16_t>().swap(p.med_cid16); raw_vector<uint8_t>().swap(p.med_cid8);
    dropc(p.irr_val); dropi(p.irr_cid); dropc(p.short_val); dropi(p.short_cid); dropc(p.rt_val); dropi(p.rt_cid);
    dropc(p.lcb.val); raw_vector<uint16_t>().swap(p.lcb.lcol);
    dropc(p.tp.val); raw_vector<uint16_t>().swap(p.tp.lrow); raw_vector<uint16_t>().swap(p.tp.lcol); std::vector<int>().swap(p.tp.dst);
    p.host_dropped = true;
    for (auto &h : p.panels) if (int rc = dasp_plan_drop_host(h.get())) return rc;
    return DASP_OK;
}

int dasp_plan_set_stream_policy(dasp_plan_t *plan, int policy)
{
    if (!plan) return DASP_ERR_ARG;
    return set_stream_policy(plan->impl, policy);
}

int dasp_plan_spmv(dasp_plan_t *plan, const void *dX, void *dY, void *stream)
{
    if (!plan) return DASP_ERR_ARG;
    return launch_spmv(plan->impl, dX, dY, stream, false);
}

int dasp_plan_spmv_acc(dasp_plan_t *plan, const void *dX, void *dY, void *stream)
{
    if (!plan) return DASP_ERR_ARG;
    return launch_spmv(plan->impl, dX, dY, stream, true);
}

int dasp_plan_time(dasp_plan_t *plan, const void *dX, void *dY, void *stream, int warmup, int iters, double *wall_ms,
                   double *event_ms)
{
    if (!plan || iters <= 0 || warmup < 0) return DASP_ERR_ARG;
    return guarded("dasp_plan_time", [&] { return time_spmv(plan->impl, dX, dY, stream, warmup, iters, wall_ms, event_ms); });
}

int dasp_plan_time_graph(dasp_plan_t *plan, const void *dX, void *dY, void *stream, int warmup, int iters, int batch,
                         double *wall_ms, double *event_ms)
{
    if (!plan || iters <= 0 || warmup < 0 || batch <= 0) return DASP_ERR_ARG;
    return guarded("dasp_plan_time_graph", [&] { return time_spmv_graph(plan->impl, dX, dY, stream, warmup, iters, batch, wall_ms, event_ms); });
}

int dasp_plan_time_each(dasp_plan_t *plan, const void *dX, void *dY, void *stream, int warmup, int iters, float *ms_each)
{
    if (!plan || iters <= 0 || warmup < 0 || !ms_each) return DASP_ERR_ARG;
    return guarded("dasp_plan_time_each", [&] { return time_spmv_each(plan->impl, dX, dY, stream, warmup, iters, ms_each); });
}

int dasp_selftest_mfma(void) { return selftest_mfma(); }

int dasp_partition_rows(int rowA, const int *rp, int n_parts, int *bounds)
{
    if (rowA < 0 || !rp || n_parts <= 0 || !bounds) return DASP_ERR_ARG;
    const long long total = rp[rowA];
    bounds[0] = 0;
    for (int g = 1; g < n_parts; ++g) {
        const long long target = total * g / n_parts;
        // first row whose start is >= target, never going backwards
        const int *it = std::lower_bound(rp, rp + rowA + 1, (int)std::min<long long>(target, 2147483647LL));
        int b = (int)(it - rp);
        b = std::max(b, bounds[g - 1]);
        bounds[g] = std::min(b, rowA);
    }
    bounds[n_parts] = rowA;
    return DASP_OK;
}

// ---- one-shot spmv_all (src/dasp_f64.h:486-1483): host buffers in/out, reference stdout + CSV
static int spmv_all_impl(int precision, const char *filename, const void *val, const int *rp, const int *ci, const void *X,
                         void *Y, int *order_rid, int rowA, int colA, int nnzA, double threshold, int block_longest)
{
    if (!X || !Y || !order_rid) { set_error("spmv_all: null X/Y/order_rid"); return DASP_ERR_ARG; }
    dasp_options_t opt;
    dasp_options_default(&opt);
    opt.threshold = threshold; opt.block_longest = block_longest;
    dasp_plan_t *plan = nullptr;
    int rc = dasp_plan_create(&plan, precision, rowA, colA, nnzA, rp, ci, val, &opt);
    if (rc) return rc;
    rc = dasp_plan_upload(plan);
    if (rc) { dasp_plan_destroy(plan); return rc; }
    const size_t vb = precision == 64 ? 8 : 2;
    void *dX = nullptr, *dY = nullptr;
    auto fail = [&](int code, const char *what) {
        set_error(what);
        if (dX) (void)hipFree(dX);
        if (dY) (void)hipFree(dY);
        dasp_plan_destroy(plan);
        return code;
    };
    if (hipMalloc(&dX, std::max<size_t>(vb * (size_t)colA, 8)) != hipSuccess) return fail(DASP_ERR_HIP, "hipMalloc X");
    if (hipMalloc(&dY, std::max<size_t>(vb * (size_t)rowA, 8)) != hipSuccess) return fail(DASP_ERR_HIP, "hipMalloc Y");
    if (hipMemcpy(dX, X, vb * (size_t)colA, hipMemcpyHostToDevice) != hipSuccess) return fail(DASP_ERR_HIP, "hipMemcpy X");
    if (hipMemset(dY, 0, vb * (size_t)rowA) != hipSuccess) return fail(DASP_ERR_HIP, "hipMemset Y");
    // f64: the reference times only its bypass kernel dasp_spmv2 (dasp_f64.h:1289-1320); f16: dasp_spmv (plain loads) then
    // dasp_spmv2 (bypass) back to back (dasp_f16.h:1557-1597).  Same here, with the plan's two cache policies.
    double wall = 0, ev = 0, wall2 = 0, ev2 = 0;
    if (precision == 16) {
        (void)dasp_plan_set_stream_policy(plan, 1);
        rc = dasp_plan_time(plan, dX, dY, nullptr, 100, 1000, &wall, &ev);
        if (rc) { std::string keep = dasp_last_error(); return fail(rc, keep.c_str()); }
    }
    (void)dasp_plan_set_stream_policy(plan, 2);
    rc = dasp_plan_time(plan, dX, dY, nullptr, 100, 1000, &wall2, &ev2);   // dasp_f64.h:1285-1286
    if (rc) { std::string keep = dasp_last_error(); return fail(rc, keep.c_str()); }
    if (precision == 64) { wall = wall2; ev = ev2; }
    if (hipMemcpy(Y, dY, vb * (size_t)rowA, hipMemcpyDeviceToHost) != hipSuccess) return fail(DASP_ERR_HIP, "hipMemcpy Y");
    std::memcpy(order_rid, dasp_plan_order(plan), sizeof(int) * (size_t)rowA);

    dasp_stats_t s;
    dasp_plan_stats(plan, &s);
    const double t = wall;                                                  // ms per SpMV, dasp_f64.h:1394
    const double gflops = (double)((long long)nnzA * 2) / (t * 1e6);        // :1395
    // the stdout line: what this run moved (the native packed bytes); the CSV row: the REFERENCE's geometry on this input, i.e. the values
    // the CUDA reference writes for the same matrix, so that the two rows can be compared column by column (time / rate columns aside)
    const long long data_X2 = s.data_X + (long long)(nnzA - colA) * (long long)vb;  // x counted per gather, :1168-1172
    const double bw1 = (double)s.data_X / (t * 1e6), bw2 = (double)data_X2 / (t * 1e6);
    std::printf("SpMV_X:  %8.4lf ms, %8.4lf GFlop/s, %9.4lf GB/s, %9.4lf GB/s\n", t, gflops, bw1, bw2);     // dasp_f64.h:1398
    const double t2 = wall2, gflops2 = (double)((long long)nnzA * 2) / (t2 * 1e6);
    if (precision == 16)                                                                                      // dasp_f16.h:1718
        std::printf("SpMV_X2: %8.4lf ms, %8.4lf GFlop/s, %9.4lf GB/s, %9.4lf GB/s\n", t2, gflops2, bw1, bw2);
    std::printf("\n");
    std::fflush(stdout);
    struct stat st;
    if (stat("data", &st) == 0 && S_ISDIR(st.st_mode)) {
        const long long ref_X2 = s.ref_data_X + (long long)(nnzA - colA) * (long long)vb;
        const double tb = precision == 16 ? t2 : t;                          // dasp_f16.h:1715-1716 rates its bandwidths on the bypass time
        const double rbw1 = (double)s.ref_data_X / (tb * 1e6), rbw2 = (double)ref_X2 / (tb * 1e6);
        FILE *fo = std::fopen(precision == 64 ? "data/spmv_f64_record.csv" : "data/spmv_f16_record.csv", "a");
        if (fo) {   // column order of dasp_f64.h:1439-1441 (f16 adds dasp_pre after rate_fill0.. see dasp_f16.h:1756-1758)
            std::fprintf(fo, "%s,%d,%d,%d,%d,%d,%d,%d,%d,%d,%d,%d,%lld,%d,%lld,%d,%lld,%d,", filename ? filename : "", rowA, colA,
                         nnzA, s.short_row_1, s.common_13, s.short_row_3, s.short_row_4, s.short_row_2, s.row_long, s.row_block,
                         s.nnz_short, s.ref_fill0_nnz_short, s.nnz_long, s.ref_fill0_nnz_long, s.ref_origin_nnz_reg, s.ref_fill0_nnz_reg, s.ref_nnz_irreg);
            if (precision == 64)
                std::fprintf(fo, "%lf,%d,%lld,%lf,%lf,%lf,%lf,", s.ref_rate_fill0, block_longest, s.ref_data_X, t, gflops, rbw1, rbw2);
            else   // dasp_f16.h:1757-1758: ..., dasp_pre, dasp_time, dasp_gflops, dasp_time_bypass, dasp_gflops_bypass, bandwidth1, bandwidth2
                std::fprintf(fo, "%lf,%d,%lld,%lf,%lf,%lf,%lf,%lf,%lf,%lf,", s.ref_rate_fill0, block_longest, s.ref_data_X, s.pre_ms, t, gflops, t2,
                             gflops2, rbw1, rbw2);
            std::fclose(fo);
        }
        // this build's own geometry, one complete line per call: name, padded sizes, rate_fill0, packed bytes, the two rates on them
        fo = std::fopen(precision == 64 ? "data/dasp_amd_native_f64.csv" : "data/dasp_amd_native_f16.csv", "a");
        if (fo) {
            std::fprintf(fo, "%s,%lld,%lld,%d,%lld,%d,%lf,%lld,%lf,%lf\n", filename ? filename : "", s.fill0_nnz_short, s.fill0_nnz_long, s.origin_nnz_reg,
                         s.fill0_nnz_reg, s.nnz_irreg, s.rate_fill0, s.data_X, bw1, bw2);
            std::fclose(fo);
        }
    }
    (void)hipFree(dX);
    (void)hipFree(dY);
    dasp_plan_destroy(plan);
    return DASP_OK;
}

int dasp_spmv_all_f64(const char *filename, const double *csrValA, const int *csrRowPtrA, const int *csrColIdxA,
                      const double *X_val, double *Y_val, int *order_rid, int rowA, int colA, int nnzA, int NUM,
                      double threshold, int block_longest)
{
    (void)NUM;
    return guarded("dasp_spmv_all_f64", [&] {
        return spmv_all_impl(64, filename, csrValA, csrRowPtrA, csrColIdxA, X_val, Y_val, order_rid, rowA, colA, nnzA, threshold, block_longest);
    });
}

int dasp_spmv_all_f16(const char *filename, const uint16_t *csrValA, const int *csrRowPtrA, const int *csrColIdxA,
                      const uint16_t *X_val, uint16_t *Y_val, int *order_rid, int rowA, int colA, int nnzA, int NUM,
                      double threshold, int block_longest)
{
    (void)NUM;
    return guarded("dasp_spmv_all_f16", [&] {
        return spmv_all_impl(16, filename, csrValA, csrRowPtrA, csrColIdxA, X_val, Y_val, order_rid, rowA, colA, nnzA, threshold, block_longest);
    });
}

}  // extern "C"

// upload.cpp -- a packed plan's way onto the device: one arena for every array (dasp_f64.h:1239-1278 does one cudaMalloc + cudaMemcpy per
// array), the launch geometry that goes with it, and the placement trials (profiles/r03_placement.md).  Host code only; the kernels it
// asks about live in kernels.hip (spmv_kernel_* below).
#include <hip/hip_runtime_api.h>

#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <type_traits>
#include <vector>

#include "plan.hpp"
#include "device.hpp"

namespace dasp {

#define HIP_TRY(expr)                                                                          \
    do {                                                                                       \
        hipError_t e_ = (expr);                                                                \
        if (e_ != hipSuccess) {                                                                \
            set_error(std::string(#expr) + ": " + hipGetErrorString(e_));                      \
            return e_ == hipErrorNoDevice ? DASP_ERR_NO_DEVICE : DASP_ERR_HIP;                 \
        }                                                                                      \
    } while (0)

Plan::~Plan()
{
    if (dev) {
        if (dev->arena) (void)hipFree(dev->arena);
        if (dev->dargs) (void)hipFree(dev->dargs);
        std::free(dev->args_sent);
        delete dev;
    }
}

int require_device()
{
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess || n <= 0) {
        set_error("no HIP device visible (the DASP GPU path has no CPU fallback)");
        return DASP_ERR_NO_DEVICE;
    }
    return DASP_OK;
}

int sync_dev_args(Plan &p)
{
    DevicePlan *d = p.dev;
    if (!d) { set_error("plan not uploaded"); return DASP_ERR_STATE; }
    if (d->dargs && d->args_sent && std::memcmp(d->args_sent, &d->args, sizeof(DevArgs)) == 0) return DASP_OK;
    if (!d->dargs) HIP_TRY(hipMalloc(&d->dargs, (sizeof(DevArgs) + 255) & ~size_t(255)));
    if (!d->args_sent) { d->args_sent = std::malloc(sizeof(DevArgs)); if (!d->args_sent) { set_error("out of host memory"); return DASP_ERR_NOMEM; } }
    HIP_TRY(hipMemcpy(d->dargs, &d->args, sizeof(DevArgs), hipMemcpyHostToDevice));
    std::memcpy(d->args_sent, &d->args, sizeof(DevArgs));
    return DASP_OK;
}

// column-blocked long rows (Plan::lcb) of a column-panel parent or of a two-phase plan (the f16 hybrid): where their arrays sit in the plan's arena, the copies, the device
// view
struct LcbOffsets { size_t v = 0, c = 0, p = 0, u = 0, d = 0, s = 0; };
template <class Place>
static LcbOffsets place_long_cb(const LongCB &L, size_t vbytes, Place &&place)
{
    LcbOffsets o;
    o.v = place(L.elems * vbytes); o.c = place(L.elems * 2); o.p = place(L.ptr.size() * 4); o.u = place(L.unit.size() * 4);
    o.d = place(L.row_dst.size() * 4); o.s = place((size_t)L.n_cb * (size_t)L.n_rows() * 8);
    return o;
}
static int upload_long_cb(Plan &p, DevicePlan *d, const LcbOffsets &o)
{
    const LongCB &L = p.lcb;
    const size_t vbytes = (size_t)p.geo.vbytes;
    char *base = static_cast<char *>(d->arena);
    HIP_TRY(hipMemcpy(base + o.v, L.val.data(), L.elems * vbytes, hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(base + o.c, L.lcol.data(), L.elems * 2, hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(base + o.p, L.ptr.data(), L.ptr.size() * 4, hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(base + o.u, L.unit.data(), L.unit.size() * 4, hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(base + o.d, L.row_dst.data(), L.row_dst.size() * 4, hipMemcpyHostToDevice));
    HIP_TRY(hipMemset(base + o.s, 0, std::max<size_t>((size_t)L.n_cb * (size_t)L.n_rows() * 8, 16)));      // empty (row, block) pieces never write their partial sum
    LcbDev &q = d->lcb;
    q.val = base + o.v; q.lcol = (const unsigned short *)(base + o.c); q.ptr = (const int *)(base + o.p); q.unit = (const int *)(base + o.u);
    q.row_dst = (const int *)(base + o.d); q.partial = base + o.s;
    q.n_units = L.n_units(); q.n_rows = L.n_rows(); q.n_cb = L.n_cb; q.cb = L.cb; q.xlen = p.n;
    if (int rc = tp_kernels_allow_lds()) return rc;
    return DASP_OK;
}

// DevicePlan::long16 from the pieces' narrow flags (host-built plans: at upload; device-built ones: again once the device packer has sent the flags back)
void choose_long16(Plan &p)
{
    DevicePlan *d = p.dev;
    if (!d) return;
    d->long16 = false;
    if (!p.windowed && p.cnt_reg8 == 0 && p.piece_c16.size() == 2 * p.piece_dst.size() && !p.piece_dst.empty()) {
        long long narrow = 0;
        for (size_t q = 0; q < p.piece_dst.size(); ++q) if (p.piece_c16[2 * q + 1]) narrow += p.piece_ptr[q + 1] - p.piece_ptr[q];
        d->long16 = narrow * 20 >= (long long)p.nnz && narrow > 0;
        if (const char *e = std::getenv("DASP_LONG16")) d->long16 = std::atoi(e) != 0;      // A/B knob
    }
}

int upload_plan(Plan &p);
static int upload_plan_impl(Plan &p)
{
    if (int rc = require_device()) return rc;
    if (p.host_dropped && p.dev) return DASP_OK;   // already on the device (packed there, or host copies released)
    if (p.host_dropped) { set_error("host arrays were dropped"); return DASP_ERR_STATE; }
    if (p.dev) { if (p.dev->arena) (void)hipFree(p.dev->arena); if (p.dev->dargs) (void)hipFree(p.dev->dargs); std::free(p.dev->args_sent); delete p.dev; p.dev = nullptr; }
    auto *d = new DevicePlan();
    p.dev = d;
    HIP_TRY(hipGetDevice(&d->device));
    if (!p.panels.empty()) {   // column panels: every panel is a plan of its own; this one only owns their partial results
        for (auto &h : p.panels) if (int rc = upload_plan(h->impl)) return rc;
        d->ypart_stride = ((size_t)std::max(p.m, 1) + 127) & ~size_t(127);
        const size_t part_bytes = d->ypart_stride * p.panels.size() * (size_t)p.geo.vbytes;
        // column-blocked long rows (Plan::lcb): their streams, tables and partial sums behind the partial-result buffers
        const LongCB &L = p.lcb;
        const size_t vbytes = (size_t)p.geo.vbytes;
        size_t total = (part_bytes + 255) & ~size_t(255);
        auto place = [&](size_t bytes) { const size_t off = total; total += (std::max<size_t>(bytes, 16) + 255) & ~size_t(255); return off; };
        const bool lcb = L.n_rows() > 0;
        LcbOffsets lo;
        if (lcb) lo = place_long_cb(L, vbytes, place);
        d->arena_bytes = total;
        HIP_TRY(hipMalloc(&d->arena, d->arena_bytes));
        HIP_TRY(hipMemset(d->arena, 0, d->arena_bytes));      // the panels never store the rows that are empty in them (DevArgs::skip0); empty (row, block) pieces never write their partial sum
        if (lcb) if (int rc = upload_long_cb(p, d, lo)) return rc;
        return DASP_OK;
    }

    if (p.two_phase) {
        // two-phase form: the tile streams + the xs stream between the two kernels, one allocation
        const TwoPhase &t = p.tp;
        const size_t S = t.segments, vbytes = (size_t)p.geo.vbytes;
        struct Item { const void *src; size_t bytes; size_t off; };
        std::vector<Item> items;
        size_t total = 0;
        auto add = [&](const void *src, size_t bytes) { const size_t off = total; items.push_back({src, bytes, off}); total += (std::max<size_t>(bytes, 16) + 255) & ~size_t(255); return off; };
        const size_t o_lc = add(t.lcol.data(), S * kTpSeg * 2), o_dst = add(t.dst.data(), S * 4), o_un = add(t.unit.data(), t.unit.size() * 4);
        const size_t o_v = add(t.val.data(), S * kTpSeg * vbytes), o_lr = add(t.lrow.data(), S * kTpSeg * 2), o_xs = add(nullptr, S * kTpSeg * vbytes);
        const size_t o_r0 = add(t.rb_row0.data(), t.rb_row0.size() * 4), o_s0 = add(t.rb_seg0.data(), t.rb_seg0.size() * 4);
        const bool lcb = p.lcb.n_rows() > 0;          // the hybrid: hub rows column-blocked beside the streams
        LcbOffsets lo;
        if (lcb) lo = place_long_cb(p.lcb, vbytes, [&](size_t bytes) { return add(nullptr, bytes); });
        HIP_TRY(hipMalloc(&d->arena, total));
        d->arena_bytes = total;
        char *base = static_cast<char *>(d->arena);
        for (const Item &it : items)
            if (it.src && it.bytes) HIP_TRY(hipMemcpy(base + it.off, it.src, it.bytes, hipMemcpyHostToDevice));
        HIP_TRY(hipMemset(base + o_xs, 0, std::max<size_t>(S * kTpSeg * vbytes, 16)));
        TpDev &q = d->tp;
        q.lcol = (const unsigned short *)(base + o_lc); q.dst = (const int *)(base + o_dst); q.unit = (const int *)(base + o_un);
        q.val = base + o_v; q.lrow = (const unsigned short *)(base + o_lr); q.xs = base + o_xs;
        q.rb_row0 = (const int *)(base + o_r0); q.rb_seg0 = (const int *)(base + o_s0);
        q.n_units = t.n_units(); q.n_rb = t.n_rb(); q.cb = t.cb; q.rb_max = t.rb_max; q.xlen = p.n; q.m = p.m;
        d->nt = true;
        if (lcb) if (int rc = upload_long_cb(p, d, lo)) return rc;
        return tp_kernels_allow_lds();
    }

    std::vector<ShortDev> groups(kNumShortGroups);
    for (int g = 0; g < kNumShortGroups; ++g) {
        groups[g].len = p.grp[g].len; groups[g].count = p.grp[g].count; groups[g].tiles = p.grp[g].tiles;
        groups[g].tile0 = p.grp[g].tile0; groups[g].elem_off = p.grp[g].elem_off; groups[g].map = p.grp[g].map;
        groups[g].seg = p.grp[g].seg; groups[g].rpt = p.grp[g].rpt;
    }
    const bool natural = p.opt.y_order == DASP_Y_NATURAL;
    const size_t part_bytes = (size_t)std::max<size_t>(1, p.multi_ptr.empty() ? 0 : (size_t)p.multi_ptr.back()) * 8;

    struct Item { const void *src; size_t bytes; size_t off; };
    std::vector<Item> items;
    size_t total = 0;
    auto add = [&](const void *src, size_t bytes) {
        size_t off = total;
        items.push_back({src, bytes, off});
        total += (bytes + 255) & ~size_t(255);
        if (bytes == 0) total += 256;
        return off;
    };
    // nnz-sized arrays: sized by their element counts; a plan packed on the device has no host copy (src = nullptr)
    auto src_of = [](const auto &v) -> const void * { return v.empty() ? nullptr : v.data(); };
    const size_t vbytes = (size_t)p.geo.vbytes;
    const size_t o_lv = add(src_of(p.long_val), p.cnt_long * vbytes);
    const size_t o_lc = add(src_of(p.long_cid), p.cnt_long * 4);
    const size_t o_lc16 = add(src_of(p.long_cid16), p.cnt_long * 2);
    const size_t o_lb = add(src_of(p.long_base), p.cnt_long_chunks * 4);
    const size_t o_pc16 = add(p.piece_c16.data(), p.piece_c16.size() * 4);
    const size_t o_pp = add(p.piece_ptr.data(), p.piece_ptr.size() * 4);
    const size_t o_pd = add(p.piece_dst.data(), p.piece_dst.size() * 4);
    const size_t o_mp = add(p.multi_ptr.data(), p.multi_ptr.size() * 4);
    const size_t o_md = add(p.multi_dst.data(), p.multi_dst.size() * 4);
    const size_t o_part = add(nullptr, part_bytes);
    const size_t o_mptr = add(p.med_ptr.data(), p.med_ptr.size() * 4);
    const size_t o_mv = add(src_of(p.med_val), p.cnt_reg * vbytes);
    const size_t o_mc = add(src_of(p.med_cid), p.cid16 ? 0 : p.cnt_reg * 4);
    const size_t o_mc16 = add(src_of(p.med_cid16), p.cid16 ? (p.cnt_reg - p.cnt_reg8) * 2 : 0);
    const size_t o_mc8 = add(src_of(p.med_cid8), p.cnt_reg8);
    const size_t o_c8p = add(p.med_c8ptr.data(), p.med_c8ptr.size() * 4);
    const size_t o_mb = add(src_of(p.med_base), p.cid16 ? (size_t)p.med_ptr.back() * 4 : 0);
    const size_t o_ip = add(p.irr_ptr.data(), p.irr_ptr.size() * 4);
    // r6: the number of tail steps of every block, so that a wave knows its block's step count from two SCALAR loads (med_ptr, med_nt) and can issue the tiles' loads at once --
    // it used to read it off irr_ptr with a vector load, one memory latency in front of every block's stream (the latency-bound plans: cop20k_A two blocks per wave, webbase-1M)
    std::vector<int> med_nt((size_t)std::max(p.stats.n_med_blocks, 0), 0);
    {
        const int TK = p.precision == 64 ? 4 : 16;
        for (size_t b = 0; b < med_nt.size(); ++b) {
            const size_t r0 = b * (size_t)kMedRows;
            if (r0 + 1 < p.irr_ptr.size()) med_nt[b] = (p.irr_ptr[r0 + 1] - p.irr_ptr[r0] + TK - 1) / TK;
        }
    }
    const size_t o_mnt = add(med_nt.data(), med_nt.size() * 4);
    const size_t o_iv = add(src_of(p.irr_val), p.cnt_irr * vbytes);
    const size_t o_ic = add(src_of(p.irr_cid), p.cnt_irr * 4);
    const size_t o_mdst = add(p.med_dst.data(), p.med_dst.size() * 4);
    const size_t o_wc = add(p.win_cmin.data(), p.win_cmin.size() * 4);
    const size_t o_wl = add(p.win_len.data(), p.win_len.size() * 4);
    const size_t o_sv = add(src_of(p.short_val), p.cnt_short * vbytes);
    const size_t o_sc = add(src_of(p.short_cid), p.cnt_short * 4);
    const size_t o_g = add(groups.data(), groups.size() * sizeof(ShortDev));
    const size_t o_rv = add(src_of(p.rt_val), p.cnt_rt * vbytes);
    const size_t o_rc = add(src_of(p.rt_cid), p.cnt_rt * 4);
    const size_t o_rp = add(p.rt_ptr.data(), p.rt_ptr.size() * 4);
    const size_t o_rs = add(p.rt_start.data(), p.rt_start.size() * 2);
    const size_t o_rm = add(p.rt_mask.data(), p.rt_mask.size() * 8);
    std::vector<int> ord_mapped;   // a column panel writes row r to dst_map[r] (its parent's slot), not to r
    if (natural && !p.dst_map.empty()) { ord_mapped.resize(p.order.size()); for (size_t i = 0; i < p.order.size(); ++i) ord_mapped[i] = p.dst_map[(size_t)p.order[i]]; }
    const size_t o_ord = add(natural ? (ord_mapped.empty() ? p.order.data() : ord_mapped.data()) : nullptr, natural ? p.order.size() * 4 : 0);

    HIP_TRY(hipMalloc(&d->arena, total));
    d->arena_bytes = total;
    if (std::getenv("DASP_VERBOSE")) std::fprintf(stderr, "[dasp upload] arena %p + %zu bytes\n", d->arena, total);
    char *base = static_cast<char *>(d->arena);
    for (const Item &it : items)
        if (it.src && it.bytes) HIP_TRY(hipMemcpy(base + it.off, it.src, it.bytes, hipMemcpyHostToDevice));

    d->map.long_val = o_lv; d->map.long_cid = o_lc; d->map.med_val = o_mv; d->map.med_cid = o_mc; d->map.med_cid16 = o_mc16; d->map.med_cid8 = o_mc8;
    d->map.med_base = o_mb; d->map.irr_val = o_iv; d->map.irr_cid = o_ic; d->map.short_val = o_sv; d->map.short_cid = o_sc;
    d->map.rt_val = o_rv; d->map.rt_cid = o_rc; d->map.long_cid16 = o_lc16; d->map.long_base = o_lb; d->map.piece_c16 = o_pc16;
    DevArgs &a = d->args;
    a.long_val = base + o_lv; a.long_cid = (const int *)(base + o_lc);
    a.long_cid16 = (const unsigned short *)(base + o_lc16); a.long_base = (const int *)(base + o_lb); a.piece_c16 = (const int *)(base + o_pc16);
    a.piece_ptr = (const int *)(base + o_pp); a.piece_dst = (const int *)(base + o_pd);
    a.multi_ptr = (const int *)(base + o_mp); a.multi_dst = (const int *)(base + o_md);
    a.partial = base + o_part;
    a.n_pieces = (int)p.piece_dst.size(); a.n_multi = (int)p.multi_dst.size();
    a.med_ptr = (const int *)(base + o_mptr); a.med_val = base + o_mv; a.med_cid = (const int *)(base + o_mc);
    a.irr_ptr = (const int *)(base + o_ip); a.med_nt = (const int *)(base + o_mnt); a.irr_val = base + o_iv; a.irr_cid = (const int *)(base + o_ic);
    a.n_blocks = p.stats.n_med_blocks; a.row_block = p.n_mfma_rows; a.row_long = p.med_slot0;
    for (int g = 0; g < kNumShortGroups; ++g) a.grp_tile0[g] = p.grp[g].tile0;
    a.short_val = base + o_sv; a.short_cid = (const int *)(base + o_sc); a.groups = (const ShortDev *)(base + o_g);
    a.n_short_tiles = p.stats.n_short_tiles;
    a.order = natural ? (const int *)(base + o_ord) : nullptr;
    a.wpw = p.windowed ? std::min(16, p.row_window / kMedRows) : kWavesPerWG;
    a.wg_long = (a.n_pieces + a.wpw - 1) / a.wpw;
    a.med_cid16 = (const unsigned short *)(base + o_mc16); a.med_base = (const int *)(base + o_mb);
    a.med_cid8 = (const unsigned char *)(base + o_mc8); a.med_c8ptr = (const int *)(base + o_c8p);
    a.med_dst = (const int *)(base + o_mdst); a.win_cmin = (const int *)(base + o_wc); a.win_len = (const int *)(base + o_wl);
    a.n_windows = (int)p.win_len.size(); a.blocks_per_win = p.windowed ? p.row_window / kMedRows : 0;
    a.win_hybrid = p.win_hybrid ? 1 : 0; a.win_rel16 = p.win_rel16 ? 1 : 0; a.pair_mode = p.pair_mode;
    a.win_xcd = 1;
    a.skip0 = p.panel ? 1 : 0;
    if (const char *e = std::getenv("DASP_WIN_XCD")) a.win_xcd = std::atoi(e);      // A/B knob
    d->win1 = false;
    d->seven_waves = false;
    choose_long16(p);
    if (p.precision == 64 && !p.windowed && p.med_ptr.size() > 1 && p.irr_ptr.size() > 1) {
        long long one = 0, all = 0;
        const int K = p.geo.med_k, nbk = (int)p.med_ptr.size() - 1;
        for (int b = 0; b < nbk; ++b) {
            const int nc = p.med_ptr[(size_t)b + 1] - p.med_ptr[(size_t)b], r0 = b * kMedRows;
            const int nt = (p.irr_ptr[(size_t)r0 + 1] - p.irr_ptr[(size_t)r0] + K - 1) / K;
            all += nc + nt;
            if (med_oneshot64(nc, nt)) one += nc + nt;
        }
        // (bandwidth-bound plans only: webbase-1M f64, 43 MB in 29 us, loses 1.7 % on the 72-register build; nlpkkt160 0.4636 -> 0.4426 ms, x0.1 = 291 MB 41.6 -> 40.7 us)
        d->seven_waves = all > 0 && one * 10 >= all * 7 && p.stats.data_X > (64ll << 20);
        if (const char *e = std::getenv("DASP_SEVEN_WAVES")) d->seven_waves = std::atoi(e) != 0;      // A/B knob
    }
    if (p.windowed) {
        int cus = 256;
        hipDeviceProp_t prop;
        if (hipGetDeviceProperties(&prop, d->device) == hipSuccess && prop.multiProcessorCount > 0) cus = prop.multiProcessorCount;
        d->win1 = a.n_windows <= cus;          // (wg_med below rounds the windows up to a multiple of 8: with 249..256 windows the grid is exactly the 256 CUs)
        if (const char *e = std::getenv("DASP_WIN1")) d->win1 = d->win1 && std::atoi(e) != 0;
    }
    a.wg_med = p.windowed ? (a.n_windows + 7) / 8 * 8 : (a.n_blocks + kWavesPerWG - 1) / kWavesPerWG;      // windows: a whole number per XCD (kernel)
    // f16 blocks of uniform length: a persistent set of 7 workgroups per CU striding over the blocks amortises the per-wave
    // set-up that weighs twice as much at 2 bytes per value (nlpkkt160 f16 0.675 -> 0.739 of the roofline, Queen_4147 f16
    // 0.873 -> 0.927).  Static striding needs equal blocks: with HV15R's 2 % of 3x longer rows it loses 10 %, and in f64 it
    // loses 3-9 % everywhere, so: f16 only, no windows, longest block <= 1.25 x the mean.
    if (!p.windowed && p.precision == 16 && a.n_blocks > 256 * 7 * kWavesPerWG) {      // (the threshold: a full set on a 256-CU device)
        int longest = 0;
        for (int b = 0; b < a.n_blocks; ++b) longest = std::max(longest, p.med_ptr[(size_t)b + 1] - p.med_ptr[(size_t)b]);
        const double mean = (double)p.med_ptr[(size_t)a.n_blocks] / (double)a.n_blocks;
        if (p.cnt_irr * 8 <= p.cnt_reg && mean > 0 && (double)longest <= 1.25 * mean) {
            // as many persistent workgroups per CU as the kernel's registers let reside at once (7 at <= 72 VGPRs), on every CU of the device
            int per_cu = 7, cus = 256;
            hipDeviceProp_t prop;
            if (hipGetDeviceProperties(&prop, d->device) == hipSuccess && prop.multiProcessorCount > 0) cus = prop.multiProcessorCount;
            const int fit = spmv_kernel_f16_resident(p.cid16);
            if (fit > 0) per_cu = std::min(per_cu, fit);
            (void)hipGetLastError();
            a.wg_med = std::min(a.wg_med, cus * per_cu);
        }
    }
    // XCD-contiguous block ranges (opt-in: DASP_XCD_BLOCKS=1) for plans of EVEN blocks (FEM / stencil rows: longest block <= 4 x the
    // mean).  Work of a block = its chunks + 2 (row tables, tail); every XCD gets an eighth of it.  Measured r3 (profiles/r03_xcd_blocks.md):
    // it removes exactly the traffic it was built for -- nlpkkt160's x pulled through eight L2s, FETCH 3.03 -> 2.54 GB per SpMV, 1.03 ->
    // 0.86 x the CSR bytes -- but that traffic was Infinity-Cache hits, not HBM reads, and the time does not move (nlpkkt160 0.4051 ->
    // 0.4105 ms, HV15R 0.4274 -> 0.4227, Queen 0.5089 -> 0.5083, f16 0-1.5 % slower), so it stays off by default.
    a.xcd_on = 0;
    for (int &v : a.xcd_blk) v = 0;
    a.med_stride = a.wg_med * kWavesPerWG < a.n_blocks ? 1 : 0;      // the medium range is a persistent, striding set of workgroups (f16 only today)
    if (const char *e = std::getenv("DASP_MED_LOOP")) a.med_stride = a.med_stride || std::atoi(e) != 0;      // A/B knob
    if (!p.windowed && a.wg_med == (a.n_blocks + kWavesPerWG - 1) / kWavesPerWG && a.n_blocks >= 8 * 256 && (int)p.med_ptr.size() == a.n_blocks + 1) {
        int longest = 0;
        for (int b = 0; b < a.n_blocks; ++b) longest = std::max(longest, p.med_ptr[(size_t)b + 1] - p.med_ptr[(size_t)b]);
        const double mean = (double)p.med_ptr[(size_t)a.n_blocks] / (double)a.n_blocks;
        const char *e = std::getenv("DASP_XCD_BLOCKS");
        const bool on = e && std::atoi(e) != 0 && (double)longest <= 4.0 * std::max(mean, 1.0);
        if (on) {
            auto work_before = [&](int b) { return (long long)p.med_ptr[(size_t)b] + 2ll * b; };
            const long long total = work_before(a.n_blocks);
            int most = 0;
            for (int k = 0; k <= 8; ++k) {
                int lo = 0, hi = a.n_blocks;                               // first block whose preceding work reaches k / 8 of the total
                while (lo < hi) { const int mid = (lo + hi) / 2; if (work_before(mid) * 8 < total * k) lo = mid + 1; else hi = mid; }
                a.xcd_blk[k] = k == 8 ? a.n_blocks : lo;
            }
            for (int k = 0; k < 8; ++k) most = std::max(most, a.xcd_blk[k + 1] - a.xcd_blk[k]);
            a.xcd_on = 1;
            a.wg_med = 8 * ((most + kWavesPerWG - 1) / kWavesPerWG);
        }
    }
    // wave-segmented groups hold 64 ELEMENTS per tile: one tile per wave leaves a wave's fixed cost (arguments, group tables) and one memory round trip per 768 bytes -- a matrix of
    // short rows only ran at 0.35 of the roofline.  Such plans (f64, no windows, never the multi-GPU step's) hand kShortTpw consecutive tiles to a wave, all loads issued up front
    a.short_tpw = p.stats.short_seg && !p.windowed && p.precision == 64 ? kShortTpw : 1;
    a.n_short_waves = 0;
    for (int g = 0; g < kNumShortGroups; ++g) { a.grp_wave0[g] = a.n_short_waves; a.n_short_waves += p.grp[g].seg && a.short_tpw > 1 ? (p.grp[g].tiles + a.short_tpw - 1) / a.short_tpw : p.grp[g].tiles; }
    a.wg_short = (a.n_short_waves + a.wpw - 1) / a.wpw;
    // windowed plans: a short tile per wave in workgroups of 16 waves puts 16 waves of 64-line gathers on each of a few CUs -- on cop20k_A (104 tiles in 7 workgroups) they
    // were the last waves of the launch to exit (9.9 us against 7.2 for the median window wave).  Where the tiles are few beside the windows they are folded into the window
    // workgroups as fillers behind their blocks (spmv_body): tile t goes to window t % n_windows.  One launch-wide rule: all tiles or none.
    // which category the dispatcher starts with (r6, the f16 builds): workgroups start in index order, and what is dispatched last is the launch's tail.  The f16 short tiles
    // are the longest-lived waves of a latency-bound launch (webbase-1M f16, per-wave stamps: 4.1 us against 3.0 for a medium block) and stood at the end of the grid:
    // short tiles first, webbase-1M f16 14.68 -> 13.7 us, x4 65.3 -> 61.6, the uniform variant 15.7 -> 15.1 (f64, measured the same way: 28.9 -> 31.2 -- its short waves
    // hold four tiles each -- so the f64 builds do not carry the rotation at all; medium blocks first: HV15R x0.1 f64 40.9 -> 39.2, not taken up)
    a.wg_rot = 0;
    const bool can_rot = p.precision == 16 && !p.windowed && p.rt_mask.empty() && !p.panel;
    if (can_rot && a.wg_short > 0 && a.wg_long + a.wg_med > 0) a.wg_rot = a.wg_long + a.wg_med;
    if (const char *e = std::getenv("DASP_WG_ROT")) {      // A/B knob: 0 = the grid as stored [long | medium | short], 1 = short tiles first, 2 = medium blocks first
        const int k = std::atoi(e);
        if (can_rot) a.wg_rot = k == 1 ? a.wg_long + a.wg_med : k == 2 ? a.wg_long : 0;
    }
    a.win_tiles = 0;
    if (p.windowed && win_fold_tiles(a.n_windows, a.n_short_tiles) > 0) {
        a.win_tiles = win_fold_tiles(a.n_windows, a.n_short_tiles);
        a.wg_short = 0;
    }
    if (const char *e = std::getenv("DASP_WIN_FOLD")) if (std::atoi(e) == 0 && a.win_tiles) { a.win_tiles = 0; a.wg_short = (a.n_short_waves + a.wpw - 1) / a.wpw; }      // A/B knob
    a.rt_val = base + o_rv; a.rt_cid = (const int *)(base + o_rc); a.rt_ptr = (const int *)(base + o_rp);
    a.rt_start = (const unsigned short *)(base + o_rs); a.rt_mask = (const unsigned long long *)(base + o_rm);
    a.n_rt_tiles = (int)p.rt_mask.size(); a.wg_rt = (a.n_rt_tiles + kWavesPerWG - 1) / kWavesPerWG; a.rt_max = p.rt_max;
    // streamed-once matrix data bypasses the caches (the reference's ld.global.cs, dasp_f64.h:34-51)
    // only when it cannot stay resident in the 256 MiB Infinity Cache between two SpMVs anyway.
    d->nt = p.opt.stream_policy == 2 || (p.opt.stream_policy != 1 && p.stats.data_X > kStreamBytes);
    if (p.windowed && p.lds_bytes > 65536) {
        // more than the default 64 KiB of dynamic LDS must be requested per kernel; done here (for both cache-policy
        // variants), not in the launch path, so that dasp_plan_spmv stays free of anything a stream capture would reject
        // the attribute belongs to the kernel, not to the plan: always ask for the device maximum, or a later plan with narrower
        // windows would lower the limit under an earlier one with wider windows
        if (int rc = spmv_kernel_allow_full_lds(p.precision, p.cid16)) return rc;
    }
    return sync_dev_args(p);
}


// ---- placement trials (r3; r4: profiles/r04_placement.md).  The same arena bytes run the HBM-bound kernels at one of two speeds ~8 % apart
// depending on where the plan's allocation and the WRITTEN vector landed relative to each other (r4: it is y that decides -- the same plan is slow
// with one y and fast with another, offsets inside an allocation change nothing, and with the y stores compiled out the kernel runs 10-19 %
// faster on a slow pair: 16-67 MB of y written back under a 3-GB read stream cost 40-75 us; write-through / non-temporal stores, 512-byte
// bursts, per-XCD sequential logs and XCD-contiguous block ranges do not cure it, and on some boxes no allocation is fast).  A user-mode library
// cannot see the cause, but a caller who wants to can look: dasp_plan_tune_placement copies the arena into `trials` - 1 fresh allocations,
// times each against the caller's own x / y and keeps the fastest.  OPT-IN since r4 (it was part of every dasp_plan_upload in r3): explicit
// calls, or DASP_PLACEMENT_TRIALS=n > 1 for the upload of host-built plans.  The default of an explicit call is ONE extra allocation
// (trials = 2); none is made unless the device has room for it beside 1 GiB of slack; nothing sleeps or launches behind the trials (the
// driver wipes released VRAM in the background at ~35 GB/s and kernels run 1-3 % slower meanwhile: a caller that times right after a trial
// warms up for that long -- bench.py does).  Cheaper, and usually enough, is to try a few y vectors instead: bench.py times the plan against
// six candidates of its own and keeps the fastest (16 MB each instead of 2.9 GB).
static void rebase_args(DevArgs &a, const char *from, const char *to, size_t bytes)
{
    auto mv = [&](auto &ptr) {
        const char *q = reinterpret_cast<const char *>(ptr);
        if (q >= from && q < from + bytes) ptr = reinterpret_cast<std::remove_reference_t<decltype(ptr)>>(const_cast<char *>(to + (q - from)));
    };
    mv(a.long_val); mv(a.long_cid); mv(a.long_cid16); mv(a.long_base); mv(a.piece_c16); mv(a.piece_ptr); mv(a.piece_dst); mv(a.partial); mv(a.multi_ptr); mv(a.multi_dst);
    mv(a.med_ptr); mv(a.med_val); mv(a.med_cid); mv(a.med_cid16); mv(a.med_base); mv(a.med_cid8); mv(a.med_c8ptr);
    mv(a.irr_ptr); mv(a.med_nt); mv(a.irr_val); mv(a.irr_cid); mv(a.med_dst); mv(a.win_cmin); mv(a.win_len);
    mv(a.short_val); mv(a.short_cid); mv(a.groups); mv(a.order);
}

int tune_placement(Plan &p, int trials, const void *dX, void *dY, double *ms_first, double *ms_kept)
{
    DevicePlan *d = p.dev;
    if (ms_first) *ms_first = 0.0;
    if (ms_kept) *ms_kept = 0.0;
    if (trials <= 0) trials = 2;      // one extra allocation
    trials = std::min(trials, 8);
    if (!d || !d->arena || trials <= 1 || d->arena_bytes < (size_t(256) << 20) || !p.panels.empty() || p.panel || p.windowed || p.two_phase) return DASP_OK;
    const bool verbose = std::getenv("DASP_VERBOSE") != nullptr;
    const size_t vb = (size_t)p.geo.vbytes, bytes = d->arena_bytes;
    const size_t xlen = p.opt.n_parts > 0 ? (size_t)p.opt.n_parts * (size_t)p.opt.part_stride : (size_t)p.n;
    void *x = nullptr, *y = nullptr;                 // scratch operands unless the caller lends its own (whose placement takes part in the effect)
    const void *ux = dX; void *uy = dY;
    hipEvent_t e0 = nullptr, e1 = nullptr;
    std::vector<void *> losers;
    auto cleanup = [&] {
        for (void *q : losers) (void)hipFree(q);
        if (x) (void)hipFree(x);
        if (y) (void)hipFree(y);
        if (e0) (void)hipEventDestroy(e0);
        if (e1) (void)hipEventDestroy(e1);
        (void)hipGetLastError();
    };
    // scratch operands: zeros (the values do not matter to the stream); a failure anywhere below leaves the plan as it is
    if (!ux) { if (hipMalloc(&x, std::max<size_t>(xlen * vb, 256)) != hipSuccess || hipMemset(x, 0, std::max<size_t>(xlen * vb, 256)) != hipSuccess) { cleanup(); return DASP_OK; } ux = x; }
    if (!uy) { if (hipMalloc(&y, ((size_t)p.m + 64) * vb) != hipSuccess) { cleanup(); return DASP_OK; } uy = y; }
    if (hipEventCreate(&e0) != hipSuccess || hipEventCreate(&e1) != hipSuccess) { cleanup(); return DASP_OK; }
    auto time_it = [&](double *ms) -> bool {
        for (int i = 0; i < 2; ++i) if (launch_spmv(p, ux, uy, nullptr, false) != DASP_OK) return false;
        if (hipEventRecord(e0, nullptr) != hipSuccess) return false;
        const int reps = 6;
        for (int i = 0; i < reps; ++i) if (launch_spmv(p, ux, uy, nullptr, false) != DASP_OK) return false;
        float t = 0.f;
        if (hipEventRecord(e1, nullptr) != hipSuccess || hipEventSynchronize(e1) != hipSuccess || hipEventElapsedTime(&t, e0, e1) != hipSuccess) return false;
        *ms = (double)t / reps;
        return true;
    };
    double best = 0.0;
    if (!time_it(&best)) { cleanup(); return DASP_OK; }
    double lo = best, hi = best;
    if (ms_first) *ms_first = best;
    if (verbose) std::fprintf(stderr, "[dasp placement] allocation 0 at %p: %.4f ms\n", d->arena, best);
    for (int t = 1; t < trials && hi < 1.04 * lo; ++t) {
        void *na = nullptr;
        size_t mem_free = 0, mem_total = 0;
        if (hipMemGetInfo(&mem_free, &mem_total) != hipSuccess || mem_free < bytes + (size_t(1) << 30)) { (void)hipGetLastError(); break; }      // never squeeze a co-resident allocator
        if (hipMalloc(&na, bytes) != hipSuccess) { (void)hipGetLastError(); break; }
        if (hipMemcpy(na, d->arena, bytes, hipMemcpyDeviceToDevice) != hipSuccess) { (void)hipFree(na); (void)hipGetLastError(); break; }
        char *old = static_cast<char *>(d->arena);
        rebase_args(d->args, old, static_cast<char *>(na), bytes);
        d->arena = na;
        double ms = 0.0;
        const bool ok = time_it(&ms);
        if (verbose) std::fprintf(stderr, "[dasp placement] allocation %d at %p: %.4f ms\n", t, na, ok ? ms : -1.0);
        if (ok && ms < best) { best = ms; losers.push_back(old); }
        else {                                        // back to the one that was faster
            rebase_args(d->args, static_cast<char *>(na), old, bytes);
            d->arena = old;
            losers.push_back(na);
        }
        if (ok) { lo = std::min(lo, ms); hi = std::max(hi, ms); }
    }
    if (hipDeviceSynchronize() != hipSuccess) (void)hipGetLastError();
    // the device-resident DevArgs follows the arena that was kept (a losing last trial rebased `args` back behind it): re-sent HERE, with nothing captured, so that
    // the next launch_spmv -- possibly inside a caller's stream capture -- finds it in sync and copies nothing
    // (column-panel parents never get here: the trials return at once for them)
    if (int rc = sync_dev_args(p)) { cleanup(); return rc; }
    if (ms_kept) *ms_kept = best;
    // the allocations that lost go back now (the driver wipes released VRAM in the background: see the note above)
    for (void *q : losers) (void)hipFree(q);
    losers.clear();
    cleanup();
    return DASP_OK;
}

// host-built plans: trials at upload only when asked for (DASP_PLACEMENT_TRIALS=n > 1); everybody else calls dasp_plan_tune_placement -- or nothing
int upload_plan(Plan &p)
{
    const bool fresh = !(p.host_dropped && p.dev);
    if (int rc = upload_plan_impl(p)) return rc;
    int trials = 1;
    if (const char *e = std::getenv("DASP_PLACEMENT_TRIALS")) trials = std::max(1, std::min(8, std::atoi(e)));
    return fresh && trials > 1 ? tune_placement(p, trials, nullptr, nullptr, nullptr, nullptr) : DASP_OK;
}
// the arena with every O(rows) array, the nnz-sized regions left for the device packers (devpack.hip)
int upload_plan_unpacked(Plan &p) { return upload_plan_impl(p); }

}  // namespace dasp

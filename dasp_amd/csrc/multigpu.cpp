// multigpu.cpp -- row-partitioned y = A*x over the GPUs of one node, one process per GPU: the dasp_mg_* part of the C ABI
// (include/dasp_amd.h).  The reference is single-GPU (src/main_f64.cu:102-168, one device, one
// spmv_all call), so there is no reference counterpart; the design follows BASELINE.json's north_star: contiguous row
// ranges, every rank runs the complete DASP pipeline on its slice, RCCL all-gather of y over xGMI.
//
// A rank's nonzeros are split by column ownership into two DASP plans (square matrices):
//     own   : columns inside the rank's own row range -> read the rank's own padded slice of x (its previous y),
//     other : every other column                      -> read the all-gather buffer (ids remapped to its padded layout),
// so that product t+1 over the own columns runs while all-gather t is still in flight on the communication stream and only
// the (small, for banded matrices) other-column product waits for it.  RCCL is reached through dlopen: one copy per process
// (the one PyTorch already mapped, if any), and libdasp_amd.so loads on machines without it.
//
// Fused step (r3, f64 plans that qualify: mg_step_supported): ONE launch per iteration on the caller's stream.  The own-column
// workgroups come first in grid order; a bounded set of persistent workgroups waits IN THE KERNEL for the flag the exchange of the
// previous iteration sets AND for the count of finished own-column workgroups, then runs the other-column plan (y += through sc1
// loads / stores: the two-launch arithmetic); the last workgroup to finish publishes "y ready", on which a one-lane kernel at the
// head of the communication stream spins.  So the caller's stream carries back-to-back kernels and nothing else -- no kernel
// boundary between the two products, no stream wait / write packets -- and every hand-off is a plain kernel on a plain device word
// (no Beta stream-memory-operation API).  A wait that times out sets a sticky error word instead of hanging; dasp_mg_check reports
// it and drops the plan to the two-launch form.  (Tried first: both products adding into a zeroed slice with f64 atomics, fully
// concurrent -- 125 us per step against 70: agent-scope atomics are performed at the memory side.)
#include <hip/hip_runtime_api.h>
#include <hip/hip_ext.h>
#include <rccl/rccl.h>

#include <dlfcn.h>

#include <algorithm>
#include <chrono>
#include <cstdlib>
#include <cstring>
#include <exception>
#include <memory>
#include <mutex>
#include <new>
#include <string>
#include <thread>
#include <vector>

#include <unistd.h>

#include "plan.hpp"
#include "device.hpp"
#include "mgx.hpp"

using namespace dasp;

namespace dasp { int devpack_spin(void *stream, int micros, int channels); }

namespace {

struct RcclApi {
    void *handle = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId *) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t *, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*AllGather)(const void *, void *, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
    const char *(*GetErrorString)(ncclResult_t) = nullptr;
    std::string err;
};

// the process-wide RCCL: an already mapped copy first (PyTorch's wheel bundles its own librccl.so with the same soname;
// two copies in one process would each bring their own bootstrap state), then the system one
RcclApi *rccl()
{
    static RcclApi api;
    static std::once_flag once;
    std::call_once(once, [] {
        const char *forced = std::getenv("DASP_RCCL_LIB");
        void *h = nullptr;
        if (forced && *forced) h = dlopen(forced, RTLD_NOW | RTLD_GLOBAL);
        if (!h) h = dlopen("librccl.so.1", RTLD_NOW | RTLD_NOLOAD);
        if (!h) h = dlopen("librccl.so", RTLD_NOW | RTLD_NOLOAD);
        if (!h) h = dlopen("librccl.so.1", RTLD_NOW | RTLD_GLOBAL);
        if (!h) h = dlopen("/opt/rocm/lib/librccl.so.1", RTLD_NOW | RTLD_GLOBAL);
        if (!h) { const char *e = dlerror(); api.err = std::string("cannot load librccl.so.1: ") + (e ? e : "?"); return; }
        api.handle = h;
        api.GetUniqueId = reinterpret_cast<decltype(api.GetUniqueId)>(dlsym(h, "ncclGetUniqueId"));
        api.CommInitRank = reinterpret_cast<decltype(api.CommInitRank)>(dlsym(h, "ncclCommInitRank"));
        api.CommDestroy = reinterpret_cast<decltype(api.CommDestroy)>(dlsym(h, "ncclCommDestroy"));
        api.AllGather = reinterpret_cast<decltype(api.AllGather)>(dlsym(h, "ncclAllGather"));
        api.GetErrorString = reinterpret_cast<decltype(api.GetErrorString)>(dlsym(h, "ncclGetErrorString"));
        if (!api.GetUniqueId || !api.CommInitRank || !api.CommDestroy || !api.AllGather) {
            api.err = "librccl.so.1 lacks ncclGetUniqueId / ncclCommInitRank / ncclCommDestroy / ncclAllGather";
            api.handle = nullptr;
        }
    });
    return &api;
}

int rccl_fail(const char *what, ncclResult_t r)
{
    RcclApi *a = rccl();
    set_error(std::string(what) + ": " + (a->GetErrorString ? a->GetErrorString(r) : "RCCL error") + " (" + std::to_string((int)r) + ")");
    return DASP_ERR_HIP;
}

#define MG_HIP(expr)                                                                   \
    do {                                                                               \
        hipError_t e_ = (expr);                                                        \
        if (e_ != hipSuccess) {                                                        \
            set_error(std::string(#expr) + ": " + hipGetErrorString(e_));              \
            return e_ == hipErrorNoDevice ? DASP_ERR_NO_DEVICE : DASP_ERR_HIP;         \
        }                                                                              \
    } while (0)

}  // namespace

namespace dasp {
// ---- fused multi-GPU step (multigpu.cpp).  Which plans qualify: f64, uploaded, no x windows, no column panels, 16-bit ids (the one
// instantiation of the step kernel), no long row cut into several pieces (their stage 2 would run behind the launch that publishes
// "y ready").  `other` may be null (no nonzero outside the rank's own columns).
bool mg_step_supported(const Plan &own, const Plan *other)
{
    auto ok = [](const Plan &p) {
        // 16-bit ids -- or no MFMA block at all (every medium row stored as a slab: nothing reads the id planes)
        // the step kernels decode short groups as slabs only (plain_wg -> short_tile<.., SEG = YS==0> without the wave-segmented form): a plan
        // with a segmented group -- a caller's short_seg, a loaded plan file -- takes the two-launch form instead of wrong results
        return p.precision == 64 && p.dev && p.dev->arena && p.panels.empty() && !p.windowed && (p.cid16 || p.stats.n_med_blocks == 0) && p.dev->args.n_multi == 0 &&
               p.stats.short_seg == 0;
    };
    return ok(own) && (!other || ok(*other));
}

// hot_at: where the medium blocks holding a marked row go in the dispatch order -- < 0: first (dasp_mg_step_kernel: the other-column product
// waits for them); 0 .. 1: behind that fraction of the other blocks' work (dasp_mg_step2_kernel: they wait for the peers' slices, so late
// enough for those to have arrived, but BEFORE the shortest blocks, which make the better tail); 1: last
void mg_step_marks(const Plan &p, const unsigned char *has_other, std::vector<unsigned char> &mark, std::vector<int> &blk_order, double hot_at)
{
    // the launch grid of upload_plan / dasp_mg_step_kernel: [ long pieces | medium blocks through blk_order | short tiles ], 4 units per workgroup
    const int n_pieces = (int)p.piece_dst.size(), n_blocks = p.stats.n_med_blocks, n_tiles = p.stats.n_short_tiles;
    const int wg_long = (n_pieces + kWavesPerWG - 1) / kWavesPerWG, wg_med = (n_blocks + kWavesPerWG - 1) / kWavesPerWG,
              wg_short = (n_tiles + kWavesPerWG - 1) / kWavesPerWG;
    mark.assign((size_t)wg_long + wg_med + wg_short, 0);
    for (int q = 0; q < n_pieces; ++q) {
        const int dst = p.piece_dst[(size_t)q];
        if (dst >= 0 && has_other[dst]) mark[(size_t)q / kWavesPerWG] = 1;
    }
    // medium blocks holding a row the other-column plan adds to go first, everything else keeps the stored (longest-first) order
    std::vector<unsigned char> hot((size_t)n_blocks, 0);
    for (int b = 0; b < n_blocks; ++b)
        for (int i = 0; i < kMedRows && b * kMedRows + i < p.n_mfma_rows; ++i)
            if (has_other[p.order[(size_t)p.med_slot0 + (size_t)b * kMedRows + i]]) { hot[(size_t)b] = 1; break; }
    blk_order.clear(); blk_order.reserve((size_t)n_blocks);
    int n_hot = 0;
    for (int b = 0; b < n_blocks; ++b) n_hot += hot[(size_t)b];
    if (hot_at < 0) {
        for (int b = 0; b < n_blocks; ++b) if (hot[(size_t)b]) blk_order.push_back(b);
        for (int b = 0; b < n_blocks; ++b) if (!hot[(size_t)b]) blk_order.push_back(b);
        for (int q = 0; q < n_hot; ++q) mark[(size_t)wg_long + q / kWavesPerWG] = 1;
    } else {
        // work of a block = its chunks + 2 (row tables, tail); the marked blocks go behind the first n0 unmarked ones (a whole number of workgroups)
        auto work = [&](int b) { return (long long)(p.med_ptr[(size_t)b + 1] - p.med_ptr[(size_t)b]) + 2; };
        long long cold_work = 0;
        for (int b = 0; b < n_blocks; ++b) if (!hot[(size_t)b]) cold_work += work(b);
        const long long before = (long long)(std::min(1.0, hot_at) * (double)cold_work);
        int n0 = 0;
        long long acc = 0;
        std::vector<int> cold; cold.reserve((size_t)(n_blocks - n_hot));
        for (int b = 0; b < n_blocks; ++b) if (!hot[(size_t)b]) cold.push_back(b);
        while (n0 < (int)cold.size() && acc < before) acc += work(cold[(size_t)n0++]);
        n0 = std::min((int)cold.size(), (n0 + kWavesPerWG - 1) / kWavesPerWG * kWavesPerWG);
        if ((int)cold.size() - n0 < kWavesPerWG) n0 = (int)cold.size();
        for (int i = 0; i < n0; ++i) blk_order.push_back(cold[(size_t)i]);
        for (int b = 0; b < n_blocks; ++b) if (hot[(size_t)b]) blk_order.push_back(b);
        for (int i = n0; i < (int)cold.size(); ++i) blk_order.push_back(cold[(size_t)i]);
        for (int q = n0; q < n0 + n_hot; ++q) mark[(size_t)wg_long + q / kWavesPerWG] = 1;
    }
    for (int g = 0; g < kNumShortGroups; ++g) {
        const ShortGroup &G = p.grp[g];
        const int SR = G.rpt > 0 ? G.rpt : p.geo.short_rows;          // rows of one tile of this group (slab or wave-segmented)
        for (int lt = 0; lt < G.tiles; ++lt) {
            const int t = G.tile0 + lt;
            for (int tt = lt * SR; tt < std::min(G.count, (lt + 1) * SR); ++tt) {
                const int slot = g < 5 ? G.map.slot(tt) : G.map.base[0] + tt;      // short_rows / slab_rows
                if (has_other[p.order[(size_t)slot]]) { mark[(size_t)wg_long + wg_med + t / kWavesPerWG] = 1; break; }
            }
        }
    }
}
}  // namespace dasp

struct dasp_mg_plan {
    int precision = 64, world = 1, rank = 0, rowA = 0, colA = 0, stride = 0;
    bool square = false, overlap = false, uploaded = false;
    bool step2 = false;                // overlap mode 2: ONE plan in the gather-buffer layout, boundary rows last -- the step on one stream (mgstep.hip)
    std::vector<int> bounds;
    long long nnz_own = 0, nnz_other = 0;
    dasp_plan_t *own = nullptr;        // own columns (x = this rank's padded slice) -- or the whole slice when there is no split
    dasp_plan_t *other = nullptr;      // other ranks' columns (x = the gather buffer); may be absent
    // device state
    int device = -1;
    void *ys[3] = {nullptr, nullptr, nullptr};  // this rank's padded slice of y, rotating: ys[cur] = the one last written = the next x slice
    void *yg = nullptr;                // world * stride: every rank's slice; for a square matrix also the x the products read
    void *xg = nullptr;                // what the products read: yg (square) or a plain colA vector (rectangular)
    int cur = 0;
    bool pending = false;              // an all-gather into yg is in flight on `cs`
    bool pending_sig = false;          // ... and its completion was published by a stream memory operation (else: ev_g)
    bool pending_lazy = false;         // ... fused step: ev_g not recorded yet (nobody but the step kernel has asked so far)
    hipStream_t cs = nullptr;          // communication stream
    hipStream_t rs = nullptr;          // dasp_mg_reserved_stream: a compute stream that keeps off `rs_cus` CUs
    int rs_cus = 0;
    hipEvent_t ev_y = nullptr, ev_g = nullptr;
    // fused step of a rank WITHOUT other-column nonzeros: nothing in its step kernel waits for the exchange, so the caller's stream is held
    // back here instead -- ev_x[k % 3] is recorded behind exchange k, and step k + 3 (the next writer of the slice exchange k reads) waits
    // for it.  ev_x_step[i] = the exchange the event stands for (0: none since the last synchronising call)
    hipEvent_t ev_x[3] = {nullptr, nullptr, nullptr};
    uint64_t ev_x_step[3] = {0, 0, 0};
    // sticky error word of every in-kernel wait: host-mapped memory, so that every entry point can look at it without a synchronisation
    int *err_word = nullptr;
    // two-launch form: cross-stream hand-offs through events (documented acquire / release semantics).  DASP_MG_SYNC=memops opts into
    // stream memory operations (hipStreamWriteValue64 / hipStreamWaitValue64, a Beta API) on two words of SIGNAL memory, one
    // allocation each as the API documents; a memory operation that fails falls back to the event for that hand-off.
    uint64_t *sig[2] = {nullptr, nullptr};   // [0] = products of step k done, [1] = all-gather of step k done
    uint64_t step = 0, pending_step = 0;
    bool use_sig = false;
    // fused step: one zeroed block of plain device words (device.hpp kMgWord*): sharded arrival counters, gathered (step of the last
    // completed exchange), own_go, ready (step whose products are complete), error word
    char *words = nullptr;
    bool fused = false;                // dasp_mg_spmv / dasp_mg_product run the one-launch step
    bool fusable = false;              // the plans qualify for it (decided at upload)
    bool gather_fine = false;          // yg is fine-grained device memory
    bool gather_coarse_asked = false;  // DASP_MG_GATHER_MEM=coarse (A/B knob): the caller takes the coarse-grained buffer knowingly
    uint64_t gathered_step = 0;        // step number of the last exchange queued on the communication stream (0: none since set_x)
    int max_pollers = 1024, poll_sleep = 1;
    int max_pollers_thin = 1024;       // ... when the exchange's kernels are thin (direct exchange: 26 registers; they fit beside any number of waiting workgroups)
    std::vector<unsigned char> mark;   // one byte per own-column workgroup of the step kernel: stores a row the other-column plan adds to
    std::vector<unsigned> mark_members;
    std::vector<int> blk_order;        // dispatch order of the own plan's medium blocks in the step kernel (marked ones first)
    void *d_blk_order = nullptr;
    int n_marked = 0, n_mark_shards = 0;
    void *d_mark = nullptr;            // device copy: mark bytes, then (256-byte aligned) the 64 per-shard counts
    long long timeout_ticks = 1000ll * 100000;  // 1 s at 100 MHz: far above any healthy wait (a first remote access may set up mappings), short against a hang
    // test hook (dasp_mg_set_fake_exchange): world > 1 without a communicator; the exchange = copies of the rank's slice into the
    // given gather buffers (this rank's own and those of peers living on the same device) + a kernel of that duration
    // direct exchange (dasp_mg_push_connect; mgx.hip): every rank stores its slice into every rank's gather buffer through peer-mapped
    // pointers and then a sequence number into the receiver's arrived[sender] word.  The gather buffer is double: exchange n fills half
    // n & 1, so that a peer's stores of exchange n + 1 never meet this rank's reads of exchange n - 1 ... n (a peer can only be one
    // exchange ahead: its next product needs this rank's slice)
    bool push = false;
    bool push_loopback = false;        // test hook: the peers are scratch memory of this rank (timing on a one-GPU box)
    void *xflags = nullptr;            // fine-grained: arrived[world] (u64), at kEpochOff: epoch_of[world] (u64)
    std::vector<void *> peer_gather, peer_flags;   // [world] base pointers (own entry: this rank's)
    std::vector<char> peer_opened;     // [world] 1: opened with hipIpcOpenMemHandle (closed in the destructor)
    void *d_push_dst = nullptr;        // device: MgPushDst[2][world], one table per half
    void *d_push_count = nullptr;      // device: unsigned[world]
    void *push_scratch = nullptr;      // loopback only
    uint64_t xseq = 0;                 // exchanges queued since the last dasp_mg_set_x / connect; the flags carry (epoch << 32) + xseq, the same on every
                                       // rank after that collective call, whatever a failed step left behind
    uint64_t epoch = 0;                // dasp_mg_set_x / connect count: what this rank last published to its peers' epoch_of[rank]
    bool peers_pending = false;        // the peers have not been seen at `epoch` yet (checked before the next exchange is queued)
    int push_wgs = 256;                // workgroups of the push kernel (DASP_MG_PUSH_WGS)
    // fused step + direct exchange: "y ready" is published by a one-lane kernel BEHIND the step kernel on the caller's stream (stream order = every workgroup
    // done) instead of by the last of the step's workgroups to arrive at a counter -- the workgroups nobody waits for then simply end: no wait for their
    // stores, no barrier, no atomic.  HV15R rank 1 of 8: 78.4 / 76.6 / 82.0 -> 76.9 / 76.9 / 80.9 us at a 0 / 30 / 45-us exchange, Queen_4147 95.7 / 94.9 / 94.3 ->
    // 94.1 / 92.8 / 92.5 (tools/ready_event_ab.sh; an EVENT to the communication stream instead: 80.8 / 88.6 / 100.5, its latency sits on the exchange's
    // critical path).  DASP_MG_READY_KERNEL=0: the in-kernel counter (A/B knob).
    bool ready_by_kernel = true;
    // one-stream step (step2): the plan's virtual workgroups, those without boundary rows first; products since dasp_mg_set_x (x_k2 lives in
    // half k2 & 1 of the gather buffer, this rank's y_k2 in its own slot there); the newest slice already sent to the peers
    std::vector<int> wg_list;
    int n_free2 = 0, n_marked2 = 0;
    void *d_wg_list = nullptr;
    void *d_push_dst2 = nullptr;       // device: MgPushDst[2][world - 1], the peers only (this rank's slot is written by its own product)
    void *d_push_count2 = nullptr;
    uint64_t k2 = 0, pushed_upto = 0;
    // grid position of the two-plan fused step's waiting workgroups, as a fraction of the own-column workgroups (DASP_MG_POLL_AT).  At the very end (r3)
    // the other-column product was a serial tail of ~10 us; at 0.8 it runs in the shadow of the own-column product: HV15R rank 3 of 8
    // 76.2 -> 73.4 us, Queen_4147 93.2 -> 90.5 (2 waiting workgroups per CU; profiles/r04_multi_gpu_step.md)
    double poll_at = 0.8;
    bool step2_pollers = false;        // DASP_MG_STEP2_POLLERS=1 (A/B knob): the bounded persistent workgroups also with one rank per device
    bool shared_device = false;        // DASP_MG_SHARED_DEVICE_RANKS: several ranks on this device (tests): the waiting workgroups are bounded
    int fake_us = -1;
    int fake_channels = 0;             // > 0: the stand-in kernel has the footprint of RCCL's (devpack.hip k_spin_fat), that many workgroups
    std::vector<void *> fake_peers;
    ncclComm_t comm = nullptr;

    size_t vb() const { return precision == 64 ? 8 : 2; }
    size_t all_bytes() const { return (size_t)stride * vb() * (size_t)world; }
    // the half of the gather buffer the last exchange filled (push mode; RCCL and the test hook use half 0 only)
    bool one_stream() const { return step2 && fused && push; }
    char *gcur() const { return static_cast<char *>(yg) + (one_stream() ? (size_t)(k2 & 1) * all_bytes() : push ? (size_t)(xseq & 1) * all_bytes() : 0); }
    int rows() const { return bounds[(size_t)rank + 1] - bounds[(size_t)rank]; }
    ~dasp_mg_plan()
    {
        if (device >= 0) (void)hipSetDevice(device);
        if (cs) (void)hipStreamSynchronize(cs);
        if (device >= 0) (void)hipDeviceSynchronize();
        if (comm && rccl()->CommDestroy) (void)rccl()->CommDestroy(comm);
        for (size_t r = 0; r < peer_opened.size(); ++r)
            if (peer_opened[r]) { if (peer_gather[r]) (void)hipIpcCloseMemHandle(peer_gather[r]); if (peer_flags[r]) (void)hipIpcCloseMemHandle(peer_flags[r]); }
        for (void *p : {xflags, d_push_dst, d_push_count, push_scratch, d_wg_list, d_push_dst2, d_push_count2}) if (p) (void)hipFree(p);
        if (ev_y) (void)hipEventDestroy(ev_y);
        if (ev_g) (void)hipEventDestroy(ev_g);
        for (hipEvent_t e : ev_x) if (e) (void)hipEventDestroy(e);
        if (err_word) (void)hipHostFree(err_word);
        if (cs) (void)hipStreamDestroy(cs);
        if (rs) (void)hipStreamDestroy(rs);
        for (uint64_t *w : sig) if (w) (void)hipFree(w);
        if (words) (void)hipFree(words);
        if (d_mark) (void)hipFree(d_mark);
        if (d_blk_order) (void)hipFree(d_blk_order);
        for (void *p : {ys[0], yg}) if (p) (void)hipFree(p);       // ys[1], ys[2] are parts of ys[0]'s allocation
        if (xg && xg != yg) (void)hipFree(xg);
        if (own) dasp_plan_destroy(own);
        if (other) dasp_plan_destroy(other);
    }
};

namespace {

// CSR slice -> (own, other): entries with lo <= col < hi (columns re-based to 0) and the rest (global columns); the order
// inside a row is kept.  Row-parallel count + scatter.
template <class T>
void split_by_owner(int m, const int *rp, const int *ci, const T *val, int lo, int hi, int threads,
                    std::vector<int> &rpO, std::vector<int> &ciO, std::vector<T> &vO,
                    std::vector<int> &rpR, std::vector<int> &ciR, std::vector<T> &vR)
{
    rpO.assign((size_t)m + 1, 0); rpR.assign((size_t)m + 1, 0);
    auto par = [&](auto f) {
        const int parts = std::max(1, std::min(threads, m / 4096 + 1));
        std::vector<std::thread> th;
        for (int t = 0; t < parts; ++t) {
            const int b = (int)((long long)m * t / parts), e = (int)((long long)m * (t + 1) / parts);
            th.emplace_back([=] { f(b, e); });
        }
        for (auto &x : th) x.join();
    };
    par([&](int b, int e) {
        for (int i = b; i < e; ++i) {
            int o = 0;
            for (int j = rp[i]; j < rp[i + 1]; ++j) o += ci[j] >= lo && ci[j] < hi;
            rpO[(size_t)i + 1] = o; rpR[(size_t)i + 1] = rp[i + 1] - rp[i] - o;
        }
    });
    for (int i = 0; i < m; ++i) { rpO[(size_t)i + 1] += rpO[i]; rpR[(size_t)i + 1] += rpR[i]; }
    ciO.resize((size_t)rpO[m]); vO.resize((size_t)rpO[m]); ciR.resize((size_t)rpR[m]); vR.resize((size_t)rpR[m]);
    par([&](int b, int e) {
        for (int i = b; i < e; ++i) {
            int o = rpO[i], r = rpR[i];
            for (int j = rp[i]; j < rp[i + 1]; ++j) {
                if (ci[j] >= lo && ci[j] < hi) { ciO[(size_t)o] = ci[j] - lo; vO[(size_t)o++] = val[j]; }
                else { ciR[(size_t)r] = ci[j]; vR[(size_t)r++] = val[j]; }
            }
        }
    });
}

template <class T>
int create_impl(dasp_mg_plan &g, const int *rp, const int *ci, const T *val, const dasp_options_t *user)
{
    const int m = g.rows();
    const int lo = g.bounds[(size_t)g.rank], hi = g.bounds[(size_t)g.rank + 1];
    dasp_options_t opt;
    if (user) opt = *user; else dasp_options_default(&opt);
    opt.y_order = DASP_Y_NATURAL;              // the padded y slice is what the all-gather sends: un-permute fused into the stores
    opt.short_seg = -1;                        // short rows as slabs: the step kernels are compiled without the wave-segmented path (registers)
    opt.two_phase = -1;                        // (the two-phase form has no column remap and no accumulate-into-the-step: DASP kernels only)
    opt.n_parts = 0; opt.part_bounds = nullptr; opt.part_stride = 0;
    const int nnz = rp[m];
    if (!g.square) {
        // rectangular: x is not y; the products read a plain colA vector, the all-gather only assembles y
        g.nnz_own = nnz; g.nnz_other = 0;
        return dasp_plan_create(&g.own, g.precision, m, g.colA, nnz, rp, ci, val, &opt);
    }
    dasp_options_t part = opt;
    part.n_parts = g.world; part.part_bounds = g.bounds.data(); part.part_stride = g.stride;
    if (!g.overlap) {
        g.nnz_own = nnz; g.nnz_other = 0;
        return dasp_plan_create(&g.own, g.precision, m, g.colA, nnz, rp, ci, val, &part);
    }
    if (g.step2) {
        // ONE plan over all columns in the gather-buffer layout; which rows read another rank's columns decides the dispatch order only
        std::vector<unsigned char> has((size_t)std::max(m, 1), 0);
        long long other = 0;
        for (int i = 0; i < m; ++i) {
            int o = 0;
            for (int j = rp[i]; j < rp[i + 1]; ++j) o += !(ci[j] >= lo && ci[j] < hi);
            has[(size_t)i] = o > 0; other += o;
        }
        g.nnz_own = nnz - other; g.nnz_other = other;
        if (int rc = dasp_plan_create(&g.own, g.precision, m, g.colA, nnz, rp, ci, val, &part)) return rc;
        const Plan &P = g.own->impl;
        if (g.precision == 64 && P.panels.empty() && !P.windowed && P.opt.y_order == DASP_Y_NATURAL) {
            // grid order: the unmarked workgroups in the plan's order, with ALL marked ones (long pieces, medium blocks, short tiles) where the
            // marked medium blocks stand -- behind `hot_at` of the other blocks' work (DASP_MG_HOT_AT; default 1.0 = last: measured with an
            // emulated link time, an earlier place gains nothing when the slices are early and loses 9-14 us when they take 30-45 us: profiles/r04_multi_gpu_step.md)
            double hot_at = 1.0;
            if (const char *e = std::getenv("DASP_MG_HOT_AT")) hot_at = std::max(0.0, std::min(1.0, std::atof(e)));
            mg_step_marks(P, has.data(), g.mark, g.blk_order, hot_at);
            const int n_pieces = (int)P.piece_dst.size(), wg_long = (n_pieces + kWavesPerWG - 1) / kWavesPerWG;
            const int wg_med = (P.stats.n_med_blocks + kWavesPerWG - 1) / kWavesPerWG;
            int first_hot_med = wg_long + wg_med;                 // the first marked medium workgroup (none: marked ones go behind the medium range)
            for (int w = wg_long; w < wg_long + wg_med; ++w) if (g.mark[(size_t)w]) { first_hot_med = w; break; }
            g.wg_list.clear(); g.wg_list.reserve(g.mark.size());
            for (int w = 0; w < first_hot_med; ++w) if (!g.mark[(size_t)w]) g.wg_list.push_back(w);
            g.n_free2 = (int)g.wg_list.size();                     // = the position of the first marked workgroup in the list
            for (size_t w = 0; w < g.mark.size(); ++w) if (g.mark[w]) g.wg_list.push_back((int)w);
            g.n_marked2 = (int)g.wg_list.size() - g.n_free2;
            for (int w = first_hot_med; w < (int)g.mark.size(); ++w) if (!g.mark[(size_t)w]) g.wg_list.push_back(w);
        }
        return DASP_OK;
    }
    std::vector<int> rpO, ciO, rpR, ciR;
    std::vector<T> vO, vR;
    split_by_owner<T>(m, rp, ci, val, lo, hi, resolve_threads(opt.host_threads), rpO, ciO, vO, rpR, ciR, vR);
    g.nnz_own = rpO[(size_t)m]; g.nnz_other = rpR[(size_t)m];
    static const int zero = 0;
    if (int rc = dasp_plan_create(&g.own, g.precision, m, g.stride, rpO[(size_t)m], rpO.data(), ciO.empty() ? &zero : ciO.data(),
                                  vO.empty() ? static_cast<const void *>(&zero) : vO.data(), &opt)) return rc;
    // the fused step runs ONE kernel instantiation for both plans (16-bit ids): where the own-column plan chose them, the small
    // other-column plan follows (left to itself it would stay below the size from which 16-bit ids are automatic)
    if (part.cid16 == 0 && g.precision == 64 && g.own->impl.cid16) part.cid16 = 1;
    if (g.nnz_other > 0)
        if (int rc = dasp_plan_create(&g.other, g.precision, m, g.colA, rpR[(size_t)m], rpR.data(), ciR.data(), vR.data(), &part)) return rc;
    // fused step: which own-column workgroups store a row the other-column plan adds to (the only ones it has to wait for)
    if (g.other && g.precision == 64 && g.own->impl.panels.empty() && !g.own->impl.windowed && g.own->impl.opt.y_order == DASP_Y_NATURAL) {
        std::vector<unsigned char> has((size_t)std::max(m, 1), 0);
        for (int i = 0; i < m; ++i) has[(size_t)i] = rpR[(size_t)i + 1] > rpR[(size_t)i];
        mg_step_marks(g.own->impl, has.data(), g.mark, g.blk_order);
        g.mark_members.assign(64, 0);
        for (size_t w = 0; w < g.mark.size(); ++w) if (g.mark[w]) { ++g.mark_members[w & 63]; ++g.n_marked; }
        for (unsigned c : g.mark_members) g.n_mark_shards += c > 0;
    }
    return DASP_OK;
}

// hand-offs of the two-launch form: events, or -- opted in -- stream memory operations; an operation that fails is replaced by the
// event (or, on the consumer side of a value that was already written by an operation, by a host wait for the communication stream).
constexpr uint64_t kAllBits = 0xFFFFFFFFFFFFFFFFull;
constexpr int kExchangeKernelRegs = 288;      // registers per lane RCCL's collective kernels need on gfx950 (see dasp_mg_upload)
// "y ready": the communication stream continues once the products queued so far on `s` are done (both sides are queued here)
int handoff_ready(dasp_mg_plan &g, hipStream_t s, uint64_t k)
{
    if (g.use_sig) {
        if (hipStreamWriteValue64(s, g.sig[0], k, 0) == hipSuccess &&
            hipStreamWaitValue64(g.cs, g.sig[0], k, hipStreamWaitValueGte, kAllBits) == hipSuccess) return DASP_OK;
        (void)hipGetLastError(); g.use_sig = false;
    }
    MG_HIP(hipEventRecord(g.ev_y, s));
    MG_HIP(hipStreamWaitEvent(g.cs, g.ev_y, 0));
    return DASP_OK;
}
// "gathered", producer side (behind the exchange on the communication stream)
int publish_gathered(dasp_mg_plan &g, uint64_t k)
{
    g.pending_sig = false;
    if (g.use_sig) {
        if (hipStreamWriteValue64(g.cs, g.sig[1], k, 0) == hipSuccess) { g.pending_sig = true; return DASP_OK; }
        (void)hipGetLastError(); g.use_sig = false;
    }
    MG_HIP(hipEventRecord(g.ev_g, g.cs));
    return DASP_OK;
}
// "gathered", consumer side on stream s.  One dasp_mg_plan is driven from ONE stream (header): the flag is cleared once that
// stream has been ordered behind the exchange.
int wait_gathered(dasp_mg_plan &g, hipStream_t s)
{
    if (!g.pending) return DASP_OK;
    if (g.pending_lazy) { MG_HIP(hipEventRecord(g.ev_g, g.cs)); g.pending_lazy = false; }      // the communication stream is in order: behind the exchange
    if (g.pending_sig) {
        if (hipStreamWaitValue64(s, g.sig[1], g.pending_step, hipStreamWaitValueGte, kAllBits) != hipSuccess) {
            (void)hipGetLastError(); g.use_sig = false;
            MG_HIP(hipStreamSynchronize(g.cs));          // no event was recorded for this exchange: wait for it on the host
        }
    } else MG_HIP(hipStreamWaitEvent(s, g.ev_g, 0));
    g.pending = false;
    return DASP_OK;
}

// a wait inside a kernel gave up since the last dasp_mg_check: whatever was computed since is invalid.  Read from host-mapped memory,
// no synchronisation -- every entry point that hands out results or queues more work looks here first
int sticky_error(const dasp_mg_plan &g)
{
    if (!g.err_word) return DASP_OK;
    const int e = *static_cast<volatile int *>(g.err_word);
    if (e == 0) return DASP_OK;
    set_error(std::string("multi-GPU step: an in-kernel wait timed out (") + (e == 1 ? "the products' wait for the previous exchange" : e == 2 ? "the exchange's wait for the products" : "a peer's slice did not arrive") +
              "); results since then are invalid -- dasp_mg_check, then dasp_mg_set_x");
    return DASP_ERR_STATE;
}
void forget_exchange_events(dasp_mg_plan &g) { for (uint64_t &k : g.ev_x_step) k = 0; }

// the one-launch step on stream s?  Not with RCCL between several ranks on a stream that leaves RCCL's kernels no room: they would
// not start while workgroups wait for them (DESIGN_MULTIGPU.md 5.3) -- such a call runs the two-launch form instead of timing out
bool fuse_on(const dasp_mg_plan &g, hipStream_t s)
{
    if (g.step2) return g.one_stream();
    return g.fused && (g.push || !g.comm || g.world == 1 || (g.rs && s == g.rs));
}

int product(dasp_mg_plan &g, hipStream_t s)
{
    const int cur = g.cur, nxt = (g.cur + 1) % 3;
    if (g.one_stream()) {
        // ONE launch, on this stream only: head workgroups send y_k2 (this rank's slot of the half that holds x) to the peers, the plan's
        // workgroups compute y_{k2+1} into this rank's slot of the other half, those with boundary rows behind the peers' arrival flags
        const size_t sl = (size_t)g.stride * g.vb();
        const bool send = g.world > 1 && g.k2 >= 1 && g.pushed_upto < g.k2;
        MgPushArgs pa{};
        if (send) {
            pa.src = g.gcur() + (size_t)g.rank * sl; pa.bytes = sl;
            pa.dst = static_cast<const MgPushDst *>(g.d_push_dst2) + (size_t)(g.k2 & 1) * (size_t)(g.world - 1);
            pa.n_dst = g.world - 1; pa.count = static_cast<unsigned *>(g.d_push_count2); pa.wgs = g.push_wgs;
            pa.seq = (g.epoch << 32) + g.k2; pa.timeout = g.timeout_ticks; pa.err = g.err_word;
            pa.delay_ticks = g.push_loopback && g.fake_us > 0 ? 100ll * g.fake_us : 0;      // timing probe: the links' share of the exchange
            g.pushed_upto = g.k2;
        }
        MgStep2Ctl c{};
        c.wg_list = g.d_wg_list; c.blk_order = g.d_blk_order; c.n_push = send ? g.push_wgs : 0; c.n_free = g.n_free2; c.n_marked = g.n_marked2;
        c.n_total = (int)g.wg_list.size();
        c.max_pollers = g.shared_device || g.step2_pollers ? g.max_pollers_thin : 0;      // 0: every marked workgroup waits by itself, in place
        c.arrived = g.xflags; c.world = g.world; c.rank = g.rank;
        c.need = g.world > 1 && g.k2 >= 1 ? (g.epoch << 32) + g.k2 : 0;
        c.err = g.err_word; c.timeout_ticks = g.timeout_ticks; c.poll_sleep = g.poll_sleep;
        c.fence_mode = 1; c.xcd_fenced = g.words + kMgWordXcd; c.step = g.step + 1;
        char *xin = g.gcur();
        char *yout = static_cast<char *>(g.yg) + (size_t)((g.k2 + 1) & 1) * g.all_bytes() + (size_t)g.rank * sl;
        if (int rc = launch_mg_step2(g.own->impl, xin, yout, c, pa, s)) return rc;
        ++g.k2; ++g.step;
        return DASP_OK;
    }
    if (fuse_on(g, s)) {
        if (!g.other && g.world > 1) {
            // no other-column plan: the step kernel has no waiting workgroups, i.e. nothing that orders this stream behind the exchange, while
            // y rotates through three slices -- step k + 3 overwrites the slice exchange k sends.  Hold the stream until that exchange is done
            // (two exchanges may still be in flight: the own-column product needs nothing from them)
            const uint64_t j = g.step + 1;
            if (j > 3 && g.ev_x_step[j % 3] == j - 3) MG_HIP(hipStreamWaitEvent(s, g.ev_x[j % 3], 0));
        }
        // one launch: own-column workgroups, then the persistent workgroups that wait in the kernel for exchange `gathered_step`
        MgStepCtl c{};
        c.err = g.err_word;
        c.words = g.words; c.need = g.gathered_step; c.step = g.step + 1;
        c.mark = g.d_mark; c.mark_members = static_cast<char *>(g.d_mark) + ((g.mark.size() + 255) & ~size_t(255));
        c.n_marked = g.n_marked; c.n_mark_shards = g.n_mark_shards; c.blk_order = g.d_blk_order;
        c.max_pollers = g.push ? g.max_pollers_thin : g.max_pollers; c.timeout_ticks = g.timeout_ticks; c.poll_sleep = g.poll_sleep;
        c.poll_at = g.poll_at;
        c.ready_by_event = g.ready_by_kernel && g.push ? 1 : 0;
        if (int rc = launch_mg_step(g.own->impl, g.other ? &g.other->impl : nullptr, g.ys[cur], g.gcur(), g.ys[nxt], c, s)) return rc;
    } else if (g.overlap && !g.step2) {
        // own columns: needs only this rank's slice of x, i.e. its own previous y -- no communication
        if (int rc = dasp_plan_spmv(g.own, g.ys[cur], g.ys[nxt], s)) return rc;
        if (int rc = wait_gathered(g, s)) return rc;                                         // the other ranks' x has arrived
        if (g.other) if (int rc = dasp_plan_spmv_acc(g.other, g.gcur(), g.ys[nxt], s)) return rc;   // y += (other columns) * x
    } else {
        if (int rc = wait_gathered(g, s)) return rc;
        if (int rc = dasp_plan_spmv(g.own, g.square ? g.gcur() : g.xg, g.ys[nxt], s)) return rc;
    }
    g.cur = nxt;
    ++g.step;
    return DASP_OK;
}

constexpr size_t kEpochOff = 2048, kFlagBytes = 4096;      // layout of xflags; at most 256 ranks

// push mode: the peers must have passed the dasp_mg_set_x / connect this rank has passed before this rank stores into their buffers
// (their earlier products are then complete, their x is in place).  Host-side: read this rank's epoch_of[] until every entry is there.
int push_wait_peers(dasp_mg_plan &g)
{
    if (!g.peers_pending) return DASP_OK;
    if (g.push_loopback) { g.peers_pending = false; return DASP_OK; }
    double limit = 120.0;
    if (const char *e = std::getenv("DASP_MG_BARRIER_TIMEOUT_S")) limit = std::max(0.01, std::atof(e));
    std::vector<uint64_t> seen((size_t)g.world, 0);
    const auto t0 = std::chrono::steady_clock::now();
    for (;;) {
        MG_HIP(hipMemcpy(seen.data(), static_cast<char *>(g.xflags) + kEpochOff, seen.size() * sizeof(uint64_t), hipMemcpyDeviceToHost));
        bool all = true;
        for (int r = 0; r < g.world; ++r) all = all && seen[(size_t)r] >= g.epoch;
        if (all) break;
        if (std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() > limit) {
            set_error("direct exchange: a peer did not reach this dasp_mg_set_x / dasp_mg_push_connect within " + std::to_string(limit) + " s (DASP_MG_BARRIER_TIMEOUT_S)");
            return DASP_ERR_STATE;
        }
        std::this_thread::sleep_for(std::chrono::microseconds(50));
    }
    g.peers_pending = false;
    return DASP_OK;
}

// publish "this rank has passed its n-th dasp_mg_set_x / connect" into every peer's epoch_of[rank] (plain copies: they also prove the
// peer mappings before any kernel stores through them)
int push_arrive(dasp_mg_plan &g)
{
    ++g.epoch;
    g.peers_pending = true;
    for (int r = 0; r < g.world; ++r) {
        char *f = static_cast<char *>(g.push_loopback ? g.xflags : g.peer_flags[(size_t)r]);
        const int slot = g.push_loopback ? r : g.rank;
        MG_HIP(hipMemcpy(f + kEpochOff + (size_t)slot * sizeof(uint64_t), &g.epoch, sizeof(uint64_t), hipMemcpyHostToDevice));
    }
    return DASP_OK;
}

// the direct exchange on stream q: [wait for `ready` >= ready_need] stores into every rank's gather buffer + flags, then one wave waits
// for every sender's flag [and sets `gathered` = step for the fused step's waiting workgroups]
int push_exchange(dasp_mg_plan &g, hipStream_t q, uint64_t ready_need, bool set_gathered, uint64_t step)
{
    if (int rc = push_wait_peers(g)) return rc;
    const uint64_t seq = (g.epoch << 32) + (++g.xseq);
    MgPushArgs a{};
    a.src = g.ys[g.cur]; a.bytes = (size_t)g.stride * g.vb();
    a.dst = static_cast<const MgPushDst *>(g.d_push_dst) + (size_t)(g.xseq & 1) * (size_t)g.world;
    a.n_dst = g.world;
    a.count = static_cast<unsigned *>(g.d_push_count);
    a.wgs = g.push_wgs; a.seq = seq;
    a.ready = reinterpret_cast<const unsigned long long *>(g.words + kMgWordReady); a.ready_need = ready_need;
    a.timeout = g.timeout_ticks; a.err = g.err_word;
    if (int rc = launch_mg_push(a, q)) return rc;
    if (g.push_loopback && g.fake_us > 0) if (int rc = devpack_spin(q, g.fake_us, 0)) return rc;       // timing probe: the links' share of the exchange
    return launch_mg_arrived(g.xflags, g.world, seq, set_gathered ? g.words + kMgWordGathered : nullptr, step, g.timeout_ticks, g.err_word, q);
}

// one-stream step: the newest slice travels with the NEXT product; who wants the gathered y before that (dasp_mg_wait, dasp_mg_get_y) sends it
// here -- the exchange as kernels of its own on stream q: stores + flags, then one wave that waits for every peer's flag
int flush_step2(dasp_mg_plan &g, hipStream_t q, bool force)
{
    if (!g.one_stream() || g.world <= 1 || g.k2 < 1 || (!force && g.pushed_upto >= g.k2)) return DASP_OK;
    if (int rc = push_wait_peers(g)) return rc;
    const size_t sl = (size_t)g.stride * g.vb();
    MgPushArgs a{};
    a.src = g.gcur() + (size_t)g.rank * sl; a.bytes = sl;
    a.dst = static_cast<const MgPushDst *>(g.d_push_dst2) + (size_t)(g.k2 & 1) * (size_t)(g.world - 1);
    a.n_dst = g.world - 1; a.count = static_cast<unsigned *>(g.d_push_count2); a.wgs = g.push_wgs;
    a.seq = (g.epoch << 32) + g.k2; a.timeout = g.timeout_ticks; a.err = g.err_word;
    if (int rc = launch_mg_push(a, q)) return rc;
    g.pushed_upto = g.k2;
    return launch_mg_arrived(g.xflags, g.world, a.seq, nullptr, 0, g.timeout_ticks, g.err_word, q, g.rank);
}

// the slice ys[cur] -> every rank's gather buffer on stream q (direct stores; RCCL; one rank or the test hook: local copies)
int exchange(dasp_mg_plan &g, hipStream_t q)
{
    if (g.one_stream()) return flush_step2(g, q, true);
    if (g.push) return push_exchange(g, q, 0, false, 0);
    if (g.comm) {
        const ncclResult_t r = rccl()->AllGather(g.ys[g.cur], g.yg, (size_t)g.stride, g.precision == 64 ? ncclFloat64 : ncclFloat16, g.comm, q);
        if (r != ncclSuccess) return rccl_fail("ncclAllGather", r);
        return DASP_OK;
    }
    const size_t sl = (size_t)g.stride * g.vb();
    MG_HIP(hipMemcpyAsync(static_cast<char *>(g.yg) + (size_t)g.rank * sl, g.ys[g.cur], sl, hipMemcpyDeviceToDevice, q));
    for (void *peer : g.fake_peers)
        if (peer && peer != g.yg) MG_HIP(hipMemcpyAsync(static_cast<char *>(peer) + (size_t)g.rank * sl, g.ys[g.cur], sl, hipMemcpyDeviceToDevice, q));
    if (g.world > 1) if (int rc = devpack_spin(q, g.fake_us, g.fake_channels)) return rc;
    return DASP_OK;
}

// what dasp_mg_push_export writes and dasp_mg_push_connect reads: DASP_MG_IPC_BYTES per rank
struct PushBlob {
    uint64_t magic;
    int32_t rank, world;
    int64_t pid;
    uint64_t gather, flags;            // the owner's device pointers (used as they are by peers inside the owner's process)
    uint64_t gather_bytes;
    uint64_t has_ipc;                  // 0: hipIpcGetMemHandle failed on the owner (the blob still serves peers inside the owner's process)
    hipIpcMemHandle_t hg, hf;
};
static_assert(sizeof(PushBlob) <= DASP_MG_IPC_BYTES, "DASP_MG_IPC_BYTES too small");
constexpr uint64_t kPushMagic = 0x3158504D50534144ull;       // "DASPMPX1"

// device tables + state of the push mode, once the peers' pointers are known
int push_enable(dasp_mg_plan &g)
{
    const size_t sl = (size_t)g.stride * g.vb(), all = g.all_bytes();
    std::vector<MgPushDst> tab((size_t)2 * g.world);
    for (int half = 0; half < 2; ++half)
        for (int r = 0; r < g.world; ++r) {
            MgPushDst &d = tab[(size_t)half * g.world + r];
            if (g.push_loopback) {
                // every "peer" is scratch memory (except this rank itself); the flags are this rank's own arrived[r]
                char *base = r == g.rank ? static_cast<char *>(g.yg) : static_cast<char *>(g.push_scratch);
                d.data = base + (size_t)half * all + (size_t)g.rank * sl;
                d.flag = reinterpret_cast<unsigned long long *>(g.xflags) + r;
            } else {
                d.data = static_cast<char *>(g.peer_gather[(size_t)r]) + (size_t)half * all + (size_t)g.rank * sl;
                d.flag = reinterpret_cast<unsigned long long *>(g.peer_flags[(size_t)r]) + g.rank;
            }
        }
    if (!g.d_push_dst) MG_HIP(hipMalloc(&g.d_push_dst, tab.size() * sizeof(MgPushDst)));
    MG_HIP(hipMemcpy(g.d_push_dst, tab.data(), tab.size() * sizeof(MgPushDst), hipMemcpyHostToDevice));
    if (!g.d_push_count) { MG_HIP(hipMalloc(&g.d_push_count, (size_t)g.world * sizeof(unsigned))); MG_HIP(hipMemset(g.d_push_count, 0, (size_t)g.world * sizeof(unsigned))); }
    if (g.step2 && g.world > 1) {      // the one-stream step sends to the peers only
        std::vector<MgPushDst> t2;
        for (int half = 0; half < 2; ++half)
            for (int r = 0; r < g.world; ++r) if (r != g.rank) t2.push_back(tab[(size_t)half * g.world + r]);
        if (!g.d_push_dst2) MG_HIP(hipMalloc(&g.d_push_dst2, t2.size() * sizeof(MgPushDst)));
        MG_HIP(hipMemcpy(g.d_push_dst2, t2.data(), t2.size() * sizeof(MgPushDst), hipMemcpyHostToDevice));
        if (!g.d_push_count2) { MG_HIP(hipMalloc(&g.d_push_count2, (size_t)g.world * sizeof(unsigned))); MG_HIP(hipMemset(g.d_push_count2, 0, (size_t)g.world * sizeof(unsigned))); }
    }
    if (!g.words) {        // plans that do not qualify for the fused step still need the error word
        void *p = nullptr;
        MG_HIP(hipMalloc(&p, kMgWordBytes));
        MG_HIP(hipMemset(p, 0, kMgWordBytes));
        g.words = static_cast<char *>(p);
    }
    if (const char *e = std::getenv("DASP_MG_READY_KERNEL")) g.ready_by_kernel = std::atoi(e) != 0;
    if (const char *e = std::getenv("DASP_MG_PUSH_WGS")) {
        // a power of two: the per-destination counters are never reset and publish at (count % wgs == 0), which survives the wrap of an
        // unsigned counter only when wgs divides 2^32
        int w = std::max(1, std::min(4096, std::atoi(e))), p2 = 1;
        while (p2 * 2 <= w) p2 *= 2;
        g.push_wgs = p2;
    }
    // the current x is in half 0 (RCCL and the test hook use no other), where exchange count 0 looks for it
    MG_HIP(hipDeviceSynchronize());
    g.xseq = 0; g.k2 = 0; g.pushed_upto = 0;
    g.push = true;
    g.pending = false; g.pending_sig = false; g.pending_lazy = false; g.gathered_step = 0;
    forget_exchange_events(g);
    return push_arrive(g);
}

// 0: RCCL (or the test hook), 1: direct stores (needs a connected plan).  Synchronises the device; the current x stays valid.
int set_exchange(dasp_mg_plan &g, int mode)
{
    MG_HIP(hipDeviceSynchronize());
    g.pending = false; g.pending_sig = false; g.pending_lazy = false; g.gathered_step = 0;
    forget_exchange_events(g);
    if ((mode == 1) == g.push) return DASP_OK;
    if (mode == 1) {
        if (!g.d_push_dst) { set_error("dasp_mg_push_connect first"); return DASP_ERR_STATE; }
        return push_enable(g);
    }
    const size_t all = g.all_bytes();
    if (g.gcur() != static_cast<char *>(g.yg)) { MG_HIP(hipMemcpy(g.yg, static_cast<char *>(g.yg) + all, all, hipMemcpyDeviceToDevice)); MG_HIP(hipDeviceSynchronize()); }
    g.push = false; g.xseq = 0; g.k2 = 0; g.pushed_upto = 0;
    return DASP_OK;
}

}  // namespace

extern "C" {

int dasp_mg_unique_id(void *id)
{
    if (!id) return DASP_ERR_ARG;
    RcclApi *a = rccl();
    if (!a->handle) { set_error(a->err); return DASP_ERR_STATE; }
    static_assert(sizeof(ncclUniqueId) == DASP_MG_ID_BYTES, "DASP_MG_ID_BYTES must match ncclUniqueId");
    ncclUniqueId u;
    const ncclResult_t r = a->GetUniqueId(&u);
    if (r != ncclSuccess) return rccl_fail("ncclGetUniqueId", r);
    std::memcpy(id, &u, sizeof u);
    return DASP_OK;
}

int dasp_mg_plan_create(dasp_mg_plan_t **out, int precision, int rowA, int colA, int n_gpus, int rank, const int *row_bounds,
                        const int *csrRowPtr, const int *csrColIdx, const void *csrVal, const dasp_options_t *opt, int overlap)
{
    if (!out) return DASP_ERR_ARG;
    *out = nullptr;
    if ((precision != 64 && precision != 16) || rowA < 0 || colA < 0 || n_gpus <= 0 || rank < 0 || rank >= n_gpus || !row_bounds || !csrRowPtr) {
        set_error("dasp_mg_plan_create: bad arguments"); return DASP_ERR_ARG;
    }
    if (row_bounds[0] != 0 || row_bounds[n_gpus] != rowA) { set_error("row_bounds must span [0,rowA]"); return DASP_ERR_ARG; }
    for (int g = 0; g < n_gpus; ++g)
        if (row_bounds[g + 1] < row_bounds[g]) { set_error("row_bounds not monotone"); return DASP_ERR_ARG; }
    try {
        std::unique_ptr<dasp_mg_plan> g(new dasp_mg_plan());
        g->precision = precision; g->world = n_gpus; g->rank = rank; g->rowA = rowA; g->colA = colA;
        g->bounds.assign(row_bounds, row_bounds + n_gpus + 1);
        int widest = 0;
        for (int k = 0; k < n_gpus; ++k) widest = std::max(widest, row_bounds[k + 1] - row_bounds[k]);
        g->stride = std::max(64, (widest + 63) / 64 * 64);        // equal padded slices: what ncclAllGather needs
        if ((long long)g->stride * n_gpus >= (1ll << 31)) { set_error("gathered vector exceeds 2^31 elements"); return DASP_ERR_ARG; }
        g->square = rowA == colA;
        g->overlap = g->square && overlap != 0 && n_gpus > 1;
        g->step2 = g->overlap && overlap == 2;
        const int m = g->rows();
        const int nnz = csrRowPtr[m];
        if (csrRowPtr[0] != 0 || nnz < 0 || (nnz > 0 && (!csrColIdx || !csrVal))) { set_error("dasp_mg_plan_create: bad local CSR"); return DASP_ERR_ARG; }
        for (int i = 0; i < m; ++i)      // the owner split walks the rows before dasp_plan_create gets to validate them
            if (csrRowPtr[i + 1] < csrRowPtr[i]) { set_error("dasp_mg_plan_create: row pointer not monotone"); return DASP_ERR_ARG; }
        for (long long j = 0; j < nnz; ++j)
            if ((unsigned)csrColIdx[j] >= (unsigned)colA) { set_error("column index out of range"); return DASP_ERR_ARG; }
        const int rc = precision == 64 ? create_impl<double>(*g, csrRowPtr, csrColIdx, static_cast<const double *>(csrVal), opt)
                                       : create_impl<uint16_t>(*g, csrRowPtr, csrColIdx, static_cast<const uint16_t *>(csrVal), opt);
        if (rc) return rc;
        *out = g.release();
        return DASP_OK;
    } catch (const std::bad_alloc &) { set_error("out of host memory"); return DASP_ERR_NOMEM; }
    catch (const std::exception &e) { set_error(std::string("dasp_mg_plan_create: ") + e.what()); return DASP_ERR_ARG; }
}

void dasp_mg_destroy(dasp_mg_plan_t *mg) { delete mg; }

int dasp_mg_upload(dasp_mg_plan_t *mg)
{
    if (!mg) return DASP_ERR_ARG;
    dasp_mg_plan &g = *mg;
    if (g.uploaded) return DASP_OK;
    if (int rc = dasp_plan_upload(g.own)) return rc;
    if (g.other) if (int rc = dasp_plan_upload(g.other)) return rc;
    (void)dasp_plan_drop_host(g.own);
    if (g.other) (void)dasp_plan_drop_host(g.other);
    MG_HIP(hipGetDevice(&g.device));
    const size_t vb = g.vb(), sl = (size_t)g.stride * vb, all = sl * (size_t)g.world;
    {   // the three rotating slices as ONE allocation: where x and y live relative to a plan's arena decides between two speeds of the
        // product (profiles/r03_placement.md), and three separate allocations could land in either group from one step to the next
        const size_t sl_pad = (sl + 4095) & ~size_t(4095);
        void *base = nullptr;
        MG_HIP(hipMalloc(&base, 3 * sl_pad));
        MG_HIP(hipMemset(base, 0, 3 * sl_pad));
        for (int k = 0; k < 3; ++k) g.ys[k] = static_cast<char *>(base) + (size_t)k * sl_pad;
    }
    // ... and the own-column plan's placement trials once more against these slices (dasp_plan_upload ran them with scratch operands)
    if (g.overlap && !g.step2) if (int rc = dasp_plan_tune_placement(g.own, 0, g.ys[0], g.ys[1], nullptr, nullptr)) return rc;
    {   // the gather buffer is written by the exchange -- this device's RCCL kernel or, over xGMI, a peer's stores -- WHILE the fused step's
        // kernel is already running and about to read it behind an in-kernel acquire.  Coarse-grained memory only promises visibility at
        // kernel boundaries; fine-grained memory is what the HSA memory model defines in-kernel cross-agent acquire / release on, so the
        // buffer is allocated fine-grained wherever a plan may run the fused step (DASP_MG_GATHER_MEM=coarse: A/B knob).  Only the
        // other-column product (a few per cent of the gathers) reads it.
        const char *e = std::getenv("DASP_MG_GATHER_MEM");
        g.gather_coarse_asked = e && std::strcmp(e, "coarse") == 0;
        // (every plan of a multi-GPU run: the direct exchange lets peers write this buffer whatever the precision or the step's form)
        const bool fine = g.world > 1 && !g.gather_coarse_asked;
        // two halves: the direct exchange alternates between them (struct dasp_mg_plan); RCCL uses the first only
        if (!(fine && hipExtMallocWithFlags(&g.yg, 2 * all, hipDeviceMallocFinegrained) == hipSuccess && g.yg)) {
            (void)hipGetLastError();
            MG_HIP(hipMalloc(&g.yg, 2 * all));
        } else g.gather_fine = true;
        MG_HIP(hipMemset(g.yg, 0, 2 * all));
    }
    if (g.square) g.xg = g.yg;
    else { const size_t xb = std::max<size_t>((size_t)g.colA * vb, 16); MG_HIP(hipMalloc(&g.xg, xb)); MG_HIP(hipMemset(g.xg, 0, xb)); }
    {   // the communication stream gets the highest priority: the all-gather's few workgroups must not queue behind the thousands of
        // the product that is launched at the same moment on the caller's stream (own-column product of the next iteration)
        int lo = 0, hi = 0;
        MG_HIP(hipDeviceGetStreamPriorityRange(&lo, &hi));
        MG_HIP(hipStreamCreateWithPriority(&g.cs, hipStreamNonBlocking, hi));
    }
    MG_HIP(hipEventCreateWithFlags(&g.ev_y, hipEventDisableTiming));
    MG_HIP(hipEventCreateWithFlags(&g.ev_g, hipEventDisableTiming));
    for (hipEvent_t &e : g.ev_x) MG_HIP(hipEventCreateWithFlags(&e, hipEventDisableTiming));
    {   // the sticky error word of the in-kernel waits, where the host can read it at any time
        void *p = nullptr;
        MG_HIP(hipHostMalloc(&p, 64, hipHostMallocMapped));
        std::memset(p, 0, 64);
        g.err_word = static_cast<int *>(p);
    }
    {   // two-launch form: events unless DASP_MG_SYNC=memops asks for stream memory operations on signal memory
        int can = 0;
        const char *e = std::getenv("DASP_MG_SYNC");
        if (e && std::strcmp(e, "memops") == 0 && hipDeviceGetAttribute(&can, hipDeviceAttributeCanUseStreamWaitValue, g.device) == hipSuccess && can) {
            bool ok = true;
            for (int k = 0; k < 2 && ok; ++k) {
                void *p = nullptr;
                ok = hipExtMallocWithFlags(&p, sizeof(uint64_t), hipMallocSignalMemory) == hipSuccess && p;
                if (ok) {
                    const uint64_t zero = 0;
                    g.sig[k] = static_cast<uint64_t *>(p);
                    ok = hipMemcpy(p, &zero, sizeof zero, hipMemcpyHostToDevice) == hipSuccess;
                }
            }
            g.use_sig = ok;
            if (!ok) for (uint64_t *&w : g.sig) { if (w) (void)hipFree(w); w = nullptr; }
        }
        (void)hipGetLastError();
    }
    {   // fused step: wherever the plans qualify (f64, square, split by columns), unless DASP_MG_FUSED=0
        const char *e = std::getenv("DASP_MG_FUSED");
        const bool want = !(e && std::strcmp(e, "0") == 0);
        // (in-kernel acquire of data another agent / kernel writes meanwhile: fine-grained memory, or the caller's explicit A/B choice)
        if (want && (g.gather_fine || g.gather_coarse_asked || g.world == 1) && g.overlap && mg_step_supported(g.own->impl, g.other ? &g.other->impl : nullptr) &&
            (g.step2 ? (int)g.wg_list.size() == g.own->impl.stats.n_workgroups : (!g.other || (int)g.mark.size() == g.own->impl.stats.n_workgroups))) {
            void *p = nullptr;
            MG_HIP(hipMalloc(&p, kMgWordBytes));
            MG_HIP(hipMemset(p, 0, kMgWordBytes));
            g.words = static_cast<char *>(p);
            const size_t mb = (g.mark.size() + 255) & ~size_t(255);
            MG_HIP(hipMalloc(&g.d_mark, mb + 256));
            MG_HIP(hipMemset(g.d_mark, 0, mb + 256));
            if (!g.mark.empty() && !g.step2) {      // (the one-stream step has its workgroup list instead)
                MG_HIP(hipMemcpy(g.d_mark, g.mark.data(), g.mark.size(), hipMemcpyHostToDevice));
                MG_HIP(hipMemcpy(static_cast<char *>(g.d_mark) + mb, g.mark_members.data(), 64 * sizeof(unsigned), hipMemcpyHostToDevice));
            }
            if (g.step2) {
                MG_HIP(hipMalloc(&g.d_wg_list, std::max<size_t>(g.wg_list.size(), 1) * sizeof(int)));
                if (!g.wg_list.empty()) MG_HIP(hipMemcpy(g.d_wg_list, g.wg_list.data(), g.wg_list.size() * sizeof(int), hipMemcpyHostToDevice));
            }
            if (!g.blk_order.empty()) {
                MG_HIP(hipMalloc(&g.d_blk_order, g.blk_order.size() * sizeof(int)));
                MG_HIP(hipMemcpy(g.d_blk_order, g.blk_order.data(), g.blk_order.size() * sizeof(int), hipMemcpyHostToDevice));
            }
            g.fused = true; g.fusable = true;
            hipDeviceProp_t prop;
            int cus = 256;
            if (hipGetDeviceProperties(&prop, g.device) == hipSuccess && prop.multiProcessorCount > 0) cus = prop.multiProcessorCount;
            // waiting workgroups per CU: they hold their registers until the exchange has landed, so the exchange's own kernel must fit
            // beside them on every CU or the step can only end by its timeout.  RCCL's kernels in this image (rcclGenericKernel<1|2|4>,
            // gfx950 code object of librccl.so: .vgpr_count 261-280, 256 threads = one wave per SIMD, 19.7 KB of LDS) need 288 of a SIMD's
            // 512 registers; a step-kernel wave holds at most 512 / residency of them (residency 5 at its 82 registers) -> 2 per CU.
            // (1 / 2 / 4 waiting workgroups per CU: 76 / 75 / 74 us per step, profiles/r03_multi_gpu_step.md)
            const int resident = std::max(1, mg_step_resident_per_cu() > 0 ? mg_step_resident_per_cu() : 5);
            int per_cu = std::max(1, std::min(std::min(4, resident - 1), (512 - kExchangeKernelRegs) * resident / 512));
            if (const char *q = std::getenv("DASP_MG_POLL_PER_CU")) per_cu = std::max(1, std::atoi(q));
            g.max_pollers = cus * per_cu;
            g.max_pollers_thin = std::getenv("DASP_MG_POLL_PER_CU") ? g.max_pollers : cus * std::max(1, std::min(2, resident - 1));      // standing at 0.8 of the grid (poll_at), 2 / 3 / 4 per CU: 74.2 / 76.2 / 78.4 us per step (HV15R rank 3 of 8; Queen_4147 alike within 0.5 us)
            // several ranks on ONE device (tests, a bench on a box with fewer GPUs than ranks): their waiting workgroups add up, and together
            // they must never fill a CU -- the peer they wait for runs on the same CUs
            if (const char *q = std::getenv("DASP_MG_SHARED_DEVICE_RANKS")) {
                const int n = std::max(1, std::atoi(q));
                g.shared_device = n > 1;
                g.max_pollers = std::max(1, g.max_pollers / n); g.max_pollers_thin = std::max(1, g.max_pollers_thin / n);
            }
            if (const char *q = std::getenv("DASP_MG_POLL_AT")) g.poll_at = std::max(0.0, std::min(1.0, std::atof(q)));
            if (const char *q = std::getenv("DASP_MG_STEP2_POLLERS")) g.step2_pollers = std::atoi(q) != 0;
            if (const char *q = std::getenv("DASP_MG_POLL_SLEEP")) g.poll_sleep = std::max(1, std::atoi(q));
        }
    }
    if (const char *q = std::getenv("DASP_MG_TIMEOUT_MS")) g.timeout_ticks = std::max(1, std::atoi(q)) * 100000ll;      // every in-kernel wait (fused step, direct exchange)
    MG_HIP(hipDeviceSynchronize());
    g.uploaded = true;
    return DASP_OK;
}

int dasp_mg_comm_init(dasp_mg_plan_t *mg, const void *id)
{
    if (!mg || !id) return DASP_ERR_ARG;
    if (mg->comm) { set_error("communicator already initialised"); return DASP_ERR_STATE; }
    RcclApi *a = rccl();
    if (!a->handle) { set_error(a->err); return DASP_ERR_STATE; }
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) { set_error("no HIP device visible"); return DASP_ERR_NO_DEVICE; }
    ncclUniqueId u;
    std::memcpy(&u, id, sizeof u);
    const ncclResult_t r = a->CommInitRank(&mg->comm, mg->world, u, mg->rank);
    if (r != ncclSuccess) { mg->comm = nullptr; return rccl_fail("ncclCommInitRank", r); }
    return DASP_OK;
}

int dasp_mg_set_x(dasp_mg_plan_t *mg, const void *x_host)
{
    if (!mg || !x_host) return DASP_ERR_ARG;
    dasp_mg_plan &g = *mg;
    if (!g.uploaded) { set_error("dasp_mg_upload first"); return DASP_ERR_STATE; }
    const size_t vb = g.vb();
    const char *x = static_cast<const char *>(x_host);
    MG_HIP(hipDeviceSynchronize());
    g.pending = false; g.pending_sig = false; g.pending_lazy = false; g.gathered_step = 0;
    forget_exchange_events(g);
    g.k2 = 0; g.pushed_upto = 0;
    g.xseq = 0;                             // direct exchange: x goes to half 0, the next exchange fills half 1 -- on every rank alike
    if (!g.square) { MG_HIP(hipMemcpy(g.xg, x, (size_t)g.colA * vb, hipMemcpyHostToDevice)); return g.push ? push_arrive(g) : DASP_OK; }
    try {
        std::vector<char> lay((size_t)g.world * g.stride * vb, 0);
        for (int k = 0; k < g.world; ++k)
            std::memcpy(lay.data() + (size_t)k * g.stride * vb, x + (size_t)g.bounds[(size_t)k] * vb, (size_t)(g.bounds[(size_t)k + 1] - g.bounds[(size_t)k]) * vb);
        MG_HIP(hipMemcpy(g.gcur(), lay.data(), lay.size(), hipMemcpyHostToDevice));
        MG_HIP(hipMemcpy(g.ys[g.cur], lay.data() + (size_t)g.rank * g.stride * vb, (size_t)g.stride * vb, hipMemcpyHostToDevice));
    } catch (const std::bad_alloc &) { set_error("out of host memory"); return DASP_ERR_NOMEM; }
    if (g.push) return push_arrive(g);       // direct exchange: the peers wait for this before they store into this rank's buffers again
    return DASP_OK;
}

int dasp_mg_product(dasp_mg_plan_t *mg, void *stream)
{
    if (!mg) return DASP_ERR_ARG;
    if (!mg->uploaded) { set_error("dasp_mg_upload first"); return DASP_ERR_STATE; }
    if (mg->one_stream()) {
        // one-stream step: this launch's head workgroups store the previous slice into every peer's gather buffer and its boundary workgroups
        // wait on the peers' flags -- "products only" does not exist in this mode, so the same gates as dasp_mg_spmv apply
        if (int rc = sticky_error(*mg)) return rc;
        if (int rc = push_wait_peers(*mg)) return rc;
    }
    return product(*mg, static_cast<hipStream_t>(stream));
}

int dasp_mg_spmv(dasp_mg_plan_t *mg, void *stream)
{
    if (!mg) return DASP_ERR_ARG;
    dasp_mg_plan &g = *mg;
    if (!g.uploaded) { set_error("dasp_mg_upload first"); return DASP_ERR_STATE; }
    if (!g.comm && !g.push && g.world > 1 && g.fake_us < 0) { set_error("dasp_mg_spmv needs dasp_mg_comm_init or dasp_mg_push_connect (or use dasp_mg_product with your own exchange)"); return DASP_ERR_STATE; }
    hipStream_t s = static_cast<hipStream_t>(stream);
    if (int rc = sticky_error(g)) return rc;             // a wait timed out earlier: no more work on top of invalid data
    if (g.push) if (int rc = push_wait_peers(g)) return rc;
    if (g.one_stream()) return product(g, s);      // the exchange of this step rides at the head of the next launch (or of dasp_mg_wait)
    if (int rc = product(g, s)) return rc;
    // y (this rank's padded slice) -> every rank's gather buffer, on the communication stream, behind the products
    const uint64_t k = g.step;
    if (fuse_on(g, s)) {
        // the communication stream spins (one lane) until the launch's last workgroup has published step k, exchanges, publishes k back
        if (g.push) {          // both waits are inside the exchange's two kernels
            // (a one-lane wait kernel ahead of the push instead of the wait inside it: 80.3 instead of 76.7 us at a 40-us exchange)
            if (g.ready_by_kernel) if (int rc = launch_mg_flag(g.words + kMgWordReady, k, s)) return rc;      // "y ready" = everything before it on this stream is done
            if (int rc = push_exchange(g, g.cs, k, true, k)) return rc;
        } else {
            if (int rc = launch_mg_wait(g.words + kMgWordReady, k, g.timeout_ticks, g.err_word, g.cs)) return rc;
            if (int rc = exchange(g, g.cs)) return rc;
            if (int rc = launch_mg_flag(g.words + kMgWordGathered, k, g.cs)) return rc;
        }
        if (!g.other && g.world > 1) { MG_HIP(hipEventRecord(g.ev_x[k % 3], g.cs)); g.ev_x_step[k % 3] = k; }      // product(): back-pressure of a rank without other-column nonzeros
        g.gathered_step = k; g.pending_sig = false; g.pending_lazy = true;      // consumers outside the step kernel: wait_gathered records the event when one shows up
    } else {
        if (int rc = handoff_ready(g, s, k)) return rc;
        if (int rc = exchange(g, g.cs)) return rc;
        if (int rc = publish_gathered(g, k)) return rc;
    }
    g.pending = true; g.pending_step = k;
    return DASP_OK;
}

// the exchange alone (no product): the current y slice -> every rank's gather buffer.  For timing the collective by itself
// (bench.py reports it next to the products so that a scaling line can be read: step ~ max(own product, all-gather) + other product).
int dasp_mg_allgather(dasp_mg_plan_t *mg, void *stream)
{
    if (!mg) return DASP_ERR_ARG;
    dasp_mg_plan &g = *mg;
    if (!g.uploaded) { set_error("dasp_mg_upload first"); return DASP_ERR_STATE; }
    if (!g.comm && !g.push && g.world > 1 && g.fake_us < 0) { set_error("dasp_mg_allgather needs dasp_mg_comm_init or dasp_mg_push_connect"); return DASP_ERR_STATE; }
    hipStream_t s = static_cast<hipStream_t>(stream);
    if (int rc = wait_gathered(g, s)) return rc;
    if (int rc = exchange(g, s)) return rc;
    return DASP_OK;
}

int dasp_mg_wait(dasp_mg_plan_t *mg, void *stream)
{
    if (!mg) return DASP_ERR_ARG;
    if (int rc = sticky_error(*mg)) return rc;
    if (mg->one_stream()) return flush_step2(*mg, static_cast<hipStream_t>(stream), false);
    return wait_gathered(*mg, static_cast<hipStream_t>(stream));
}

int dasp_mg_get_y(dasp_mg_plan_t *mg, void *y_host)
{
    if (!mg || !y_host) return DASP_ERR_ARG;
    dasp_mg_plan &g = *mg;
    if (!g.uploaded) { set_error("dasp_mg_upload first"); return DASP_ERR_STATE; }
    if (g.one_stream() && g.pushed_upto < g.k2) {      // the newest slice has not travelled yet: behind everything queued so far (the products may be on a non-blocking stream)
        MG_HIP(hipDeviceSynchronize());
        if (int rc = flush_step2(g, nullptr, false)) return rc;
    }
    MG_HIP(hipDeviceSynchronize());
    g.pending = false; g.pending_lazy = false;
    forget_exchange_events(g);
    if (int rc = sticky_error(g)) return rc;
    const size_t vb = g.vb();
    try {
        std::vector<char> lay((size_t)g.world * g.stride * vb);
        MG_HIP(hipMemcpy(lay.data(), g.gcur(), lay.size(), hipMemcpyDeviceToHost));
        char *y = static_cast<char *>(y_host);
        for (int k = 0; k < g.world; ++k)
            std::memcpy(y + (size_t)g.bounds[(size_t)k] * vb, lay.data() + (size_t)k * g.stride * vb, (size_t)(g.bounds[(size_t)k + 1] - g.bounds[(size_t)k]) * vb);
    } catch (const std::bad_alloc &) { set_error("out of host memory"); return DASP_ERR_NOMEM; }
    return DASP_OK;
}

int dasp_mg_get_y_local(dasp_mg_plan_t *mg, void *y_host)
{
    if (!mg || !y_host) return DASP_ERR_ARG;
    if (!mg->uploaded) { set_error("dasp_mg_upload first"); return DASP_ERR_STATE; }
    MG_HIP(hipDeviceSynchronize());
    if (int rc = sticky_error(*mg)) return rc;
    const void *src = mg->one_stream() ? static_cast<const void *>(mg->gcur() + (size_t)mg->rank * mg->stride * mg->vb()) : mg->ys[mg->cur];
    if (mg->rows() > 0) MG_HIP(hipMemcpy(y_host, src, (size_t)mg->rows() * mg->vb(), hipMemcpyDeviceToHost));
    return DASP_OK;
}

void *dasp_mg_y_local(dasp_mg_plan_t *mg)
{
    if (!mg || !mg->uploaded) return nullptr;
    return mg->one_stream() ? static_cast<void *>(mg->gcur() + (size_t)mg->rank * mg->stride * mg->vb()) : mg->ys[mg->cur];
}
void *dasp_mg_gathered(dasp_mg_plan_t *mg) { return mg && mg->uploaded ? mg->gcur() : nullptr; }
void *dasp_mg_x(dasp_mg_plan_t *mg) { return mg && mg->uploaded ? (mg->square ? static_cast<void *>(mg->gcur()) : mg->xg) : nullptr; }

dasp_plan_t *dasp_mg_subplan(dasp_mg_plan_t *mg, int which)
{
    if (!mg || which < 0 || which > 1) return nullptr;
    return which == 0 ? mg->own : mg->other;
}

int dasp_mg_info(const dasp_mg_plan_t *mg, dasp_mg_info_t *out)
{
    if (!mg || !out) return DASP_ERR_ARG;
    std::memset(out, 0, sizeof *out);
    out->precision = mg->precision; out->n_gpus = mg->world; out->rank = mg->rank; out->rowA = mg->rowA; out->colA = mg->colA;
    out->row_begin = mg->bounds[(size_t)mg->rank]; out->row_end = mg->bounds[(size_t)mg->rank + 1]; out->stride = mg->stride;
    out->nnz_own = mg->nnz_own; out->nnz_other = mg->nnz_other;
    out->overlap = mg->overlap ? 1 : 0; out->has_comm = mg->comm ? 1 : 0; out->square = mg->square ? 1 : 0;
    out->stream_memops = mg->use_sig ? 1 : 0;
    out->fused_step = mg->fused ? (mg->step2 ? 2 : 1) : 0;
    out->exchange = mg->push ? 1 : 0;
    return DASP_OK;
}

int dasp_mg_check(dasp_mg_plan_t *mg)
{
    if (!mg) return DASP_ERR_ARG;
    dasp_mg_plan &g = *mg;
    if (!g.uploaded) { set_error("dasp_mg_upload first"); return DASP_ERR_STATE; }
    MG_HIP(hipDeviceSynchronize());
    g.pending = false; g.pending_lazy = false;
    forget_exchange_events(g);
    const int err = g.err_word ? *static_cast<volatile int *>(g.err_word) : 0;
    if (err == 0) return DASP_OK;
    *static_cast<volatile int *>(g.err_word) = 0;
    if (g.words) MG_HIP(hipMemset(g.words, 0, kMgWordBytes));
    g.gathered_step = 0;
    if ((int)err == 3) {
        // direct exchange: a sender's flag did not arrive in time.  Results since then are invalid.  The exchange is NOT switched here:
        // which exchange the ranks use is a collective decision (dasp_mg_set_exchange on every rank, then dasp_mg_set_x)
        set_error("direct exchange: a peer's slice did not arrive within the time-out; results since then are invalid -- dasp_mg_set_exchange(0) on every rank for RCCL, then dasp_mg_set_x");
        return DASP_ERR_STATE;
    }
    // a poll gave up: products since then lack other-column terms (1) or an exchange ran ahead of its product (2).  Drop to the
    // two-launch form; the caller starts again from dasp_mg_set_x.
    g.fused = false;
    set_error(std::string("fused multi-GPU step: ") + ((int)err == 1 ? "the wait for the previous exchange" : "the exchange's wait for the product") +
              " timed out; the plan now runs the two-launch form -- call dasp_mg_set_x and start again");
    return DASP_ERR_STATE;
}

int dasp_mg_set_fused(dasp_mg_plan_t *mg, int on)
{
    if (!mg) return DASP_ERR_ARG;
    dasp_mg_plan &g = *mg;
    if (!g.uploaded) { set_error("dasp_mg_upload first"); return DASP_ERR_STATE; }
    MG_HIP(hipDeviceSynchronize());
    if (on && !g.fusable) { set_error("this plan does not qualify for the fused step (f64, square, column split, 16-bit ids, no windows / panels / multi-piece rows)"); return DASP_ERR_STATE; }
    g.pending = false; g.pending_lazy = false; g.gathered_step = 0;
    forget_exchange_events(g);
    g.fused = on != 0;
    return DASP_OK;
}

int dasp_mg_set_fake_exchange(dasp_mg_plan_t *mg, int micros, int n_peers, void *const *peer_gathered)
{
    if (!mg || n_peers < 0 || (n_peers > 0 && !peer_gathered)) return DASP_ERR_ARG;
    mg->fake_us = micros;
    if (const char *e = std::getenv("DASP_MG_FAKE_CHANNELS")) mg->fake_channels = std::max(0, std::atoi(e));
    mg->fake_peers.assign(peer_gathered, peer_gathered + n_peers);
    return DASP_OK;
}

int dasp_mg_push_export(dasp_mg_plan_t *mg, void *blob)
{
    if (!mg || !blob) return DASP_ERR_ARG;
    dasp_mg_plan &g = *mg;
    if (!g.uploaded) { set_error("dasp_mg_upload first"); return DASP_ERR_STATE; }
    if (!g.xflags) {
        if (g.world > (int)(kEpochOff / sizeof(uint64_t))) { set_error("direct exchange: more than 256 ranks"); return DASP_ERR_ARG; }
        MG_HIP(hipExtMallocWithFlags(&g.xflags, kFlagBytes, hipDeviceMallocFinegrained));
        MG_HIP(hipMemset(g.xflags, 0, kFlagBytes));
        MG_HIP(hipDeviceSynchronize());
    }
    PushBlob b{};
    b.magic = kPushMagic; b.rank = g.rank; b.world = g.world; b.pid = (int64_t)getpid();
    b.gather = reinterpret_cast<uint64_t>(g.yg); b.flags = reinterpret_cast<uint64_t>(g.xflags); b.gather_bytes = 2 * g.all_bytes();
    b.has_ipc = hipIpcGetMemHandle(&b.hg, g.yg) == hipSuccess && hipIpcGetMemHandle(&b.hf, g.xflags) == hipSuccess ? 1 : 0;
    if (!b.has_ipc) (void)hipGetLastError();           // e.g. HSA_ENABLE_IPC_MODE_LEGACY unset on a host that only supports dmabuf IPC: peers in other processes will be told
    std::memset(blob, 0, DASP_MG_IPC_BYTES);
    std::memcpy(blob, &b, sizeof b);
    return DASP_OK;
}

int dasp_mg_push_connect(dasp_mg_plan_t *mg, const void *blobs)
{
    if (!mg || !blobs) return DASP_ERR_ARG;
    dasp_mg_plan &g = *mg;
    if (!g.uploaded || !g.xflags) { set_error("dasp_mg_push_export first"); return DASP_ERR_STATE; }
    if (g.push) return DASP_OK;
    if (!g.gather_fine && !g.gather_coarse_asked && g.world > 1) {
        set_error("dasp_mg_push_connect: the gather buffer is not fine-grained device memory (hipExtMallocWithFlags failed at upload); peers' stores into it are only defined at kernel boundaries -- use RCCL, or set DASP_MG_GATHER_MEM=coarse to accept that");
        return DASP_ERR_STATE;
    }
    MG_HIP(hipSetDevice(g.device));
    std::vector<void *> pg((size_t)g.world, nullptr), pf((size_t)g.world, nullptr);
    std::vector<char> opened((size_t)g.world, 0);
    auto undo = [&] {
        for (int r = 0; r < g.world; ++r)
            if (opened[(size_t)r]) { if (pg[(size_t)r]) (void)hipIpcCloseMemHandle(pg[(size_t)r]); if (pf[(size_t)r]) (void)hipIpcCloseMemHandle(pf[(size_t)r]); }
        (void)hipGetLastError();
    };
    for (int r = 0; r < g.world; ++r) {
        PushBlob b;
        std::memcpy(&b, static_cast<const char *>(blobs) + (size_t)r * DASP_MG_IPC_BYTES, sizeof b);
        if (b.magic != kPushMagic || b.rank != r || b.world != g.world || b.gather_bytes != 2 * g.all_bytes()) {
            undo(); set_error("dasp_mg_push_connect: entry " + std::to_string(r) + " is not rank " + std::to_string(r) + "'s dasp_mg_push_export of a plan with this partition");
            return DASP_ERR_ARG;
        }
        if (r == g.rank) { pg[(size_t)r] = g.yg; pf[(size_t)r] = g.xflags; continue; }
        if (b.pid == (int64_t)getpid()) {      // a peer inside this process (tests; one process driving several GPUs): its pointers as they are
            pg[(size_t)r] = reinterpret_cast<void *>(b.gather); pf[(size_t)r] = reinterpret_cast<void *>(b.flags);
            continue;
        }
        if (!b.has_ipc) {
            undo(); set_error("dasp_mg_push_connect: rank " + std::to_string(r) + " could not export IPC handles (hipIpcGetMemHandle failed there; HSA_ENABLE_IPC_MODE_LEGACY=0 set?)");
            return DASP_ERR_HIP;
        }
        opened[(size_t)r] = 1;
        hipError_t e = hipIpcOpenMemHandle(&pg[(size_t)r], b.hg, hipIpcMemLazyEnablePeerAccess);
        if (e == hipSuccess) e = hipIpcOpenMemHandle(&pf[(size_t)r], b.hf, hipIpcMemLazyEnablePeerAccess);
        if (e != hipSuccess) {
            undo(); set_error(std::string("dasp_mg_push_connect: hipIpcOpenMemHandle (rank ") + std::to_string(r) + "): " + hipGetErrorString(e));
            return DASP_ERR_HIP;
        }
    }
    g.peer_gather = pg; g.peer_flags = pf; g.peer_opened = opened;
    if (int rc = push_enable(g)) { g.push = false; return rc; }
    return DASP_OK;
}

int dasp_mg_push_loopback(dasp_mg_plan_t *mg)
{
    if (!mg) return DASP_ERR_ARG;
    dasp_mg_plan &g = *mg;
    if (!g.uploaded) { set_error("dasp_mg_upload first"); return DASP_ERR_STATE; }
    if (g.push) return DASP_OK;
    if (!g.xflags) {
        MG_HIP(hipExtMallocWithFlags(&g.xflags, kFlagBytes, hipDeviceMallocFinegrained));
        MG_HIP(hipMemset(g.xflags, 0, kFlagBytes));
    }
    if (!g.push_scratch) MG_HIP(hipMalloc(&g.push_scratch, 2 * g.all_bytes()));
    g.push_loopback = true;
    return push_enable(g);
}

void *dasp_mg_reserved_stream(dasp_mg_plan_t *mg, int reserve_cus)
{
    if (!mg || reserve_cus <= 0) { set_error("dasp_mg_reserved_stream: bad arguments"); return nullptr; }
    dasp_mg_plan &g = *mg;
    if (!g.uploaded) { set_error("dasp_mg_upload first"); return nullptr; }
    if (g.rs) return g.rs;
    hipDeviceProp_t prop;
    if (hipSetDevice(g.device) != hipSuccess || hipGetDeviceProperties(&prop, g.device) != hipSuccess) { (void)hipGetLastError(); set_error("dasp_mg_reserved_stream: no device"); return nullptr; }
    // KFD deals the bits of a queue's CU mask to the XCDs in turn (bit i -> XCD i % 8) and, inside an XCD, to its four shader engines in
    // turn (tools/micro/cumask.hip: the top 32 bits are one CU of every shader engine of every XCD).  A workgroup is bound to a shader
    // engine before it looks for a CU, so the reserve has to be whole groups of 32: with the top 16 bits only, half of a 16-workgroup
    // kernel of RCCL's footprint still started 100 us late.
    const int cus = prop.multiProcessorCount, group = 32;
    const int keep_off = (reserve_cus + group - 1) / group * group;
    if (cus < 2 * group || keep_off * 2 > cus) { set_error("dasp_mg_reserved_stream: cannot keep " + std::to_string(keep_off) + " of " + std::to_string(cus) + " CUs free"); return nullptr; }
    std::vector<uint32_t> mask((size_t)(cus + 31) / 32, 0xFFFFFFFFu);
    for (int b = cus - keep_off; b < (int)mask.size() * 32; ++b) mask[(size_t)b / 32] &= ~(1u << (b % 32));
    hipStream_t s = nullptr;
    const hipError_t e = hipExtStreamCreateWithCUMask(&s, (uint32_t)mask.size(), mask.data());
    if (e != hipSuccess) { (void)hipGetLastError(); set_error(std::string("hipExtStreamCreateWithCUMask: ") + hipGetErrorString(e)); return nullptr; }
    g.rs = s; g.rs_cus = keep_off;
    return s;
}

int dasp_mg_set_exchange(dasp_mg_plan_t *mg, int mode)
{
    if (!mg || mode < 0 || mode > 1) return DASP_ERR_ARG;
    if (!mg->uploaded) { set_error("dasp_mg_upload first"); return DASP_ERR_STATE; }
    return set_exchange(*mg, mode);
}

}  // extern "C"

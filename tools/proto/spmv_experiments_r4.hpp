// MOVED OUT OF THE PRODUCT TREE in r5: the r4 experiment kernels, kept as a record of what profiles/experiments_r1_r3.md and
// profiles/r04_*.md measured.  They were compiled into kernels.hip under -DDASP_EXPERIMENT together with switchable store policies inside put_y
// (DevArgs::ymode); those hooks no longer exist in dasp_amd/csrc -- to rebuild an experiment, check out revision 056d623 (tools/build_rev.sh).
// Kernels of the experiment build (-DDASP_EXPERIMENT, tools/build_variant.sh) -- never part of the product library.  All of them are the
// non-windowed f64 16-bit-id kernel with its tables read through the CONSTANT address space (YS = 3 / 5), so that assembly stores and
// atomics do not turn the table reads into vector loads.  Results: profiles/r04_placement.md (store policies) and
// profiles/r04_resident_waves.md (resident waves with y deferred in LDS).
#pragma once

// one block per wave, stores by DevArgs::ymode (tools/store_policy_probe.py)
__global__ __launch_bounds__(256, 6) void dasp_spmv_kt_kernel(DevArgs a)
{
    plain_wg<double, true, true, true, 3>(a, (int)blockIdx.x, __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)), (int)(threadIdx.x & 63), nullptr);
}
// experiment: resident workgroups walk the virtual workgroups with the grid's stride; a wave keeps the results of its medium blocks in
// LDS and writes them to y in one burst once it has no block left (permuted y order only; n_blocks must fit the LDS given)
__global__ __launch_bounds__(256, 6) void dasp_spmv_persist_kernel(DevArgs a)
{
    extern __shared__ __attribute__((aligned(16))) char lds_raw[];
    double *ybuf = reinterpret_cast<double *>(lds_raw);
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)), lane = (int)(threadIdx.x & 63);
    const int G = (int)gridDim.x, total = a.wg_long + a.wg_med + a.wg_short;
    int k = 0, v0 = -1;
    for (int v = (int)blockIdx.x; v < total; v += G) {
        if (v >= a.wg_long && v < a.wg_long + a.wg_med) {
            const int q = (v - a.wg_long) * kWavesPerWG + wave;
            if (v0 < 0) v0 = v;
            if (q < a.n_blocks) {
                const XGlobalY<double> x{static_cast<const double *>(a.x), ybuf + (k * kWavesPerWG + wave) * 16};
                medium_block<double, true, true, 0, true, 5>(a, q, lane, x);
            }
            ++k;
        } else
            plain_wg<double, true, true, true, 3>(a, v, wave, lane, nullptr);
    }
    // the burst: four blocks of 16 rows per step
    for (int j = lane >> 4; j < k; j += 4) {
        const int q = (v0 + j * G - a.wg_long) * kWavesPerWG + wave;
        const int r = q * kMedRows + (lane & 15);
        if (q < a.n_blocks && r < a.row_block) put_y<double, 3>(a, a.row_long + r, ybuf[(j * kWavesPerWG + wave) * 16 + (lane & 15)]);
    }
}
// experiment: a hybrid grid.  The first R workgroups are resident ones that share the HEAD of the medium blocks (the first H virtual workgroups: the longest blocks)
// with the grid's stride, their results parked in LDS and written in one burst; the workgroups behind them are ordinary ones -- long pieces, the TAIL of the medium
// blocks (one block per wave, direct stores), short tiles -- which the hardware hands out as the resident ones finish: the dispatcher levels what the static
// shares left uneven.
__global__ __launch_bounds__(256, 6) void dasp_spmv_hybrid_kernel(DevArgs a, int R, int H)
{
    extern __shared__ __attribute__((aligned(16))) char lds_raw[];
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)), lane = (int)(threadIdx.x & 63);
    const int wg = (int)blockIdx.x;
    if (wg < R) {
        double *ybuf = reinterpret_cast<double *>(lds_raw);
        int k = 0;
        for (int v = wg; v < H; v += R, ++k) {
            const int q = v * kWavesPerWG + wave;
            if (q < a.n_blocks) {
                const XGlobalY<double> x{static_cast<const double *>(a.x), ybuf + (k * kWavesPerWG + wave) * 16};
                medium_block<double, true, true, 0, true, 5>(a, q, lane, x);
            }
        }
        for (int j = lane >> 4; j < k; j += 4) {
            const int q = (wg + j * R) * kWavesPerWG + wave;
            const int r = q * kMedRows + (lane & 15);
            if (q < a.n_blocks && r < a.row_block) put_y<double, 3>(a, a.row_long + r, ybuf[(j * kWavesPerWG + wave) * 16 + (lane & 15)]);
        }
        return;
    }
    // ordinary workgroups: virtual id v over [long | medium tail | short]
    int v = wg - R;
    if (v >= a.wg_long) v += H;          // skip the head of the medium range
    plain_wg<double, true, true, true, 3>(a, v, wave, lane, nullptr);
}
constexpr int kCtrStride = (4096 + 256) / 4;      // counters on different memory channels
// experiment: `reps` virtual workgroups per workgroup (grid-strided), stores as they come: between one block per wave and resident waves
__global__ __launch_bounds__(256, 6) void dasp_spmv_ktr_kernel(DevArgs a, int total)
{
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)), lane = (int)(threadIdx.x & 63);
    for (int v = (int)blockIdx.x; v < total; v += (int)gridDim.x) plain_wg<double, true, true, true, 3>(a, v, wave, lane, nullptr);
}
// experiment: as above, but every wave takes its next unit (a wave's share of a virtual workgroup) from a device counter, one unit ahead of
// the one it is working on; the results of up to `cap` medium blocks wait in LDS with their block numbers.  ctr[0]: next unit, ctr[1]: waves done
// (the last one zeroes both for the next launch).
__global__ __launch_bounds__(256, 6) void dasp_spmv_persist2_kernel(DevArgs a, unsigned *ctr, int cap, int groups)
{
    extern __shared__ __attribute__((aligned(16))) char lds_raw[];
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)), lane = (int)(threadIdx.x & 63);
    double *yb = reinterpret_cast<double *>(lds_raw) + (size_t)wave * cap * 16;
    int *qb = reinterpret_cast<int *>(lds_raw + (size_t)kWavesPerWG * cap * 128) + wave * cap;
    const unsigned total = (unsigned)(a.wg_long + a.wg_med + a.wg_short) * kWavesPerWG;
    // group g of workgroups shares counter g and owns the virtual workgroups v = g (mod groups): unit i of the group = wave i & 3 of workgroup (i >> 2) groups + g
    const unsigned g = blockIdx.x % (unsigned)groups;
    unsigned *mine = ctr + (size_t)(g + 1) * kCtrStride;
    auto grab = [&]() -> unsigned { unsigned t = 0; if (lane == 0) t = __hip_atomic_fetch_add(mine, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); return t; };
    auto unit = [&](unsigned i) -> unsigned { return ((i >> 2) * (unsigned)groups + g) * kWavesPerWG + (i & 3); };
    auto flush = [&](int k) {
        for (int j = lane >> 4; j < k; j += 4) {
            const int r = qb[j] * kMedRows + (lane & 15);
            if (r < a.row_block) put_y<double, 3>(a, a.row_long + r, yb[j * 16 + (lane & 15)]);
        }
    };
    int k = 0;
    unsigned u = unit((unsigned)__builtin_amdgcn_readfirstlane((int)grab()));
    while (u < total) {
        const unsigned next = grab();
        const int v = (int)(u >> 2), wv = (int)(u & 3);
        if (v >= a.wg_long && v < a.wg_long + a.wg_med) {
            const int q = (v - a.wg_long) * kWavesPerWG + wv;
            if (q < a.n_blocks) {
                const XGlobalY<double> x{static_cast<const double *>(a.x), yb + k * 16};
                medium_block<double, true, true, 0, true, 5>(a, q, lane, x);
                if (lane == 0) qb[k] = q;
                if (++k == cap) { flush(k); k = 0; }
            }
        } else
            plain_wg<double, true, true, true, 3>(a, v, wv, lane, nullptr);
        u = unit((unsigned)__builtin_amdgcn_readfirstlane((int)next));
    }
    flush(k);
    if (lane == 0 && __hip_atomic_fetch_add(ctr, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == gridDim.x * kWavesPerWG - 1) {
        __hip_atomic_store(ctr, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        for (int i = 1; i <= groups; ++i) __hip_atomic_store(ctr + (size_t)i * kCtrStride, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}

// 1: not an experiment launch (the caller goes on), otherwise the launch's return code
static int launch_experiment(const DevArgs &a, hipStream_t s)
{
    auto on = [](const char *name) { const char *e = std::getenv(name); return e ? std::atoi(e) : 0; };
    if (on("DASP_KT_KERNEL")) {
        DevArgs b = a; b.wg_med = (b.n_blocks + kWavesPerWG - 1) / kWavesPerWG; b.xcd_on = 0;
        hipLaunchKernelGGL(dasp_spmv_kt_kernel, dim3(b.wg_long + b.wg_med + b.wg_short), dim3(256), 0, s, b);
        HIP_TRY(hipGetLastError());
        return DASP_OK;
    }
    if (!a.order && on("DASP_PERSIST")) {
        DevArgs b = a; b.wg_med = (b.n_blocks + kWavesPerWG - 1) / kWavesPerWG; b.xcd_on = 0;
        const int per_cu = on("DASP_PERSIST");
        const int G = 256 * per_cu, kmax = (b.wg_med + G - 1) / G + 1;
        const size_t need = (size_t)kmax * kWavesPerWG * 16 * sizeof(double);
        if (need <= (size_t)160 * 1024 / per_cu - 512) {
            static bool once = false;
            if (!once) { once = true; HIP_TRY(hipFuncSetAttribute((const void *)dasp_spmv_persist_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024)); }
            hipLaunchKernelGGL(dasp_spmv_persist_kernel, dim3(G), dim3(256), need, s, b);
            HIP_TRY(hipGetLastError());
            return DASP_OK;
        }
    }
    if (!a.order && on("DASP_HYBRID")) {      // DASP_HYBRID = percent of the medium virtual workgroups in the resident head; DASP_HYBRID_PER_CU resident workgroups per CU (6)
        DevArgs b = a; b.wg_med = (b.n_blocks + kWavesPerWG - 1) / kWavesPerWG; b.xcd_on = 0;
        const int per_cu = on("DASP_HYBRID_PER_CU") > 0 ? on("DASP_HYBRID_PER_CU") : 6, R = 256 * per_cu;
        const int H = std::min(b.wg_med, (int)((long long)b.wg_med * on("DASP_HYBRID") / 100));
        const int kmax = (H + R - 1) / R + 1;
        const size_t need = (size_t)kmax * kWavesPerWG * 16 * sizeof(double);
        if (need <= (size_t)160 * 1024 / 6 - 512) {
            static bool once = false;
            if (!once) { once = true; HIP_TRY(hipFuncSetAttribute((const void *)dasp_spmv_hybrid_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024)); }
            hipLaunchKernelGGL(dasp_spmv_hybrid_kernel, dim3(R + b.wg_long + (b.wg_med - H) + b.wg_short), dim3(256), need, s, b, R, H);
            HIP_TRY(hipGetLastError());
            return DASP_OK;
        }
    }
    if (on("DASP_KT_REPS")) {
        DevArgs b = a; b.wg_med = (b.n_blocks + kWavesPerWG - 1) / kWavesPerWG; b.xcd_on = 0;
        const int reps = on("DASP_KT_REPS"), total = b.wg_long + b.wg_med + b.wg_short;
        hipLaunchKernelGGL(dasp_spmv_ktr_kernel, dim3(((total + reps - 1) / reps + 7) & ~7), dim3(256), 0, s, b, total);
        HIP_TRY(hipGetLastError());
        return DASP_OK;
    }
    if (!a.order && on("DASP_PERSIST2")) {
        DevArgs b = a; b.wg_med = (b.n_blocks + kWavesPerWG - 1) / kWavesPerWG; b.xcd_on = 0;
        const int per_cu = on("DASP_PERSIST2");
        const int cap = on("DASP_PERSIST_CAP") > 0 ? on("DASP_PERSIST_CAP") : (160 * 1024 / per_cu - 512) / (kWavesPerWG * 132);
        static unsigned *ctr = nullptr;
        if (!ctr) {
            HIP_TRY(hipMalloc(&ctr, 1100 * kCtrStride * 4)); HIP_TRY(hipMemset(ctr, 0, 1100 * kCtrStride * 4));
            HIP_TRY(hipFuncSetAttribute((const void *)dasp_spmv_persist2_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024));
        }
        hipLaunchKernelGGL(dasp_spmv_persist2_kernel, dim3(256 * per_cu), dim3(256), (size_t)cap * kWavesPerWG * 132, s, b, ctr, cap,
                           on("DASP_PERSIST_GROUPS") > 0 ? on("DASP_PERSIST_GROUPS") : 64);
        HIP_TRY(hipGetLastError());
        return DASP_OK;
    }
    return 1;
}

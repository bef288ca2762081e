// tests/san/san_stubs.cpp -- TEST INFRASTRUCTURE for the CPU sanitizer builds (make -C dasp_amd/csrc san / tsan).
// The host side of the library (plan.cpp planio.cpp mmio.cpp gen.cpp capi.cpp multigpu.cpp) is compiled with -fsanitize and linked with
// this file INSTEAD of the device objects (kernels.hip, devpack.hip, mgstep.hip, mgx.hip, upload.cpp): every device entry point answers
// "no device", exactly what the real ones answer on a machine without a GPU.  GPU AddressSanitizer is not available on this pool.
#include <string>
#include <vector>
#include <memory>

#include "../../dasp_amd/csrc/plan.hpp"
#include "../../dasp_amd/csrc/device.hpp"
#include "../../dasp_amd/csrc/mgx.hpp"

namespace dasp {
static int nodev() { set_error("sanitizer build: no device code"); return DASP_ERR_NO_DEVICE; }
Plan::~Plan() { delete dev; }
int require_device() { return nodev(); }
int upload_plan(Plan &) { return nodev(); }
int upload_plan_unpacked(Plan &) { return nodev(); }
int tune_placement(Plan &, int, const void *, void *, double *, double *) { return nodev(); }
int launch_spmv(Plan &, const void *, void *, void *, bool) { return nodev(); }
int set_stream_policy(Plan &, int) { return nodev(); }
int selftest_mfma() { return nodev(); }
int download_array(Plan &, const char *, void *, size_t) { return nodev(); }
int time_spmv(Plan &, const void *, void *, void *, int, int, double *, double *) { return nodev(); }
int time_spmv_each(Plan &, const void *, void *, void *, int, int, float *) { return nodev(); }
int time_spmv_graph(Plan &, const void *, void *, void *, int, int, int, double *, double *) { return nodev(); }
int devpack_validate(const Plan &, const DevCsr &) { return nodev(); }
int devpack_window_spans(const Plan &, const DevCsr &, const raw_vector<int> &, int, int *, int *, long long *) { return nodev(); }
int devpack_line_scatter(const Plan &, const DevCsr &, const std::vector<int> &, long long *, long long *) { return nodev(); }
int devpack_row_coherence(const Plan &, const DevCsr &, const std::vector<int> &, int, long long *, long long *) { return nodev(); }
int devpack_chunk_spans(const Plan &, const DevCsr &, const raw_vector<int> &, const raw_vector<int> &, const std::vector<int> &, int *, unsigned long long *) { return nodev(); }
int devpack_all(Plan &, const DevCsr &, const PackMeta &) { return nodev(); }
int devpack_finish_panels(Plan &) { return nodev(); }
int devpack_fetch_csr(const Plan &, const DevCsr &, int *, void *) { return nodev(); }
int devpack_current_device() { return -1; }
void devpack_use_device(int) {}
int devpack_gather_columns(const Plan &, const DevCsr &, const std::vector<long long> *, long long, long long, long long, std::vector<int> &) { return nodev(); }
int devpack_panel_split(const Plan &, const DevCsr &, const std::vector<int> &, int, std::vector<std::vector<int>> &, std::vector<DevCsr> &, std::vector<std::shared_ptr<void>> &) { return nodev(); }
int devpack_row_tiles(const Plan &, DevCsr &, const std::vector<int> &, const std::vector<int> &, size_t, std::vector<std::shared_ptr<void>> &, DevRowTiles *) { return nodev(); }
int devpack_place_row_tiles(Plan &, const DevRowTiles &) { return nodev(); }
int devpack_sort_columns(const Plan &, const DevCsr &, std::vector<std::shared_ptr<void>> &, DevCsr *, int *) { return nodev(); }
int devpack_spin(void *, int, int) { return nodev(); }
int launch_mg_step(Plan &, Plan *, const void *, const void *, void *, const MgStepCtl &, void *) { return nodev(); }
int launch_mg_step2(Plan &, const void *, void *, const MgStep2Ctl &, const MgPushArgs &, void *) { return nodev(); }
int mg_step_resident_per_cu() { return 0; }
int launch_mg_wait(const void *, unsigned long long, long long, void *, void *) { return nodev(); }
int launch_mg_flag(void *, unsigned long long, void *) { return nodev(); }
int launch_mg_push(const MgPushArgs &, void *) { return nodev(); }
int launch_mg_arrived(const void *, int, unsigned long long, void *, unsigned long long, long long, void *, void *, int) { return nodev(); }
}  // namespace dasp

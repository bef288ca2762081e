"""ctypes binding of libdasp_amd.so (include/dasp_amd.h).

The library is the product: hand-written HIP kernels + C++ host preprocessing behind a C ABI.
There is no Python or CPU fallback for the compute path; if the shared object is missing the
import fails loudly, and device entry points return DASP_ERR_NO_DEVICE without a GPU.
"""
import ctypes as C
import os
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
# DASP_AMD_SO: another build of the same library (tools/ A/B probes: dasp_amd/variants/<tag>/libdasp_amd.so); never a fallback
SO_PATH = os.environ.get("DASP_AMD_SO") or os.path.join(_HERE, "libdasp_amd.so")


class DaspError(RuntimeError):
    def __init__(self, status, msg):
        super().__init__("dasp status %d: %s" % (status, msg))
        self.status = status


class Options(C.Structure):
    _fields_ = [
        ("threshold", C.c_double), ("block_longest", C.c_int), ("y_order", C.c_int), ("long_piece", C.c_int),
        ("host_threads", C.c_int), ("n_parts", C.c_int), ("part_bounds", C.POINTER(C.c_int)), ("part_stride", C.c_int),
        ("x_window", C.c_int), ("row_window", C.c_int), ("cid16", C.c_int), ("stream_policy", C.c_int),
        ("col_panels", C.c_int), ("slab_max_len", C.c_int), ("x_window_hybrid", C.c_int), ("piece_min_len", C.c_int),
        ("chunk_pairs", C.c_int), ("cid8", C.c_int), ("short_seg", C.c_int), ("row_tile_max", C.c_int), ("sort_columns", C.c_int),
        ("two_phase", C.c_int), ("tp_col_block", C.c_int), ("tp_row_block", C.c_int), ("long_cb", C.c_int),
    ]


class Stats(C.Structure):
    _fields_ = [(n, C.c_int) for n in (
        "precision rowA colA nnzA short_row_1 common_13 short_row_3 short_row_4 short_row_2 row_long row_block "
        "row_zero nnz_short nnz_long origin_nnz_reg nnz_irreg rowloop").split()] + [
        ("fill0_nnz_short", C.c_longlong), ("fill0_nnz_long", C.c_longlong), ("fill0_nnz_reg", C.c_longlong),
        ("rate_fill0", C.c_double), ("data_X", C.c_longlong), ("data_origin1", C.c_longlong),
        ("n_med_blocks", C.c_int), ("n_long_pieces", C.c_int), ("n_long_multi", C.c_int), ("n_short_tiles", C.c_int),
        ("n_workgroups", C.c_int), ("pre_ms", C.c_double),
        ("x_window_on", C.c_int), ("n_windows", C.c_int), ("n_windows_lds", C.c_int), ("lds_bytes", C.c_int),
        ("row_window", C.c_int), ("window_nnz_frac", C.c_double), ("cid16_on", C.c_int),
        ("n_col_panels", C.c_int), ("x_window_hybrid", C.c_int), ("med_rows_as_pieces", C.c_int), ("chunk_pairs", C.c_int), ("cid8_chunks", C.c_int), ("short_seg", C.c_int),
        ("row_tile_max", C.c_int), ("n_row_tiles", C.c_int), ("row_tile_nnz", C.c_longlong),
        ("two_phase", C.c_int), ("tp_col_block", C.c_int), ("tp_row_blocks", C.c_int), ("tp_units", C.c_int), ("tp_segments", C.c_longlong), ("tp_seg_elems", C.c_int),
        ("lcb_rows", C.c_int), ("lcb_col_block", C.c_int), ("lcb_units", C.c_int), ("lcb_elems", C.c_longlong),
        ("ref_fill0_nnz_short", C.c_longlong), ("ref_fill0_nnz_long", C.c_longlong), ("ref_fill0_nnz_reg", C.c_longlong), ("ref_data_X", C.c_longlong),
        ("ref_nnz_irreg", C.c_int), ("ref_origin_nnz_reg", C.c_int), ("ref_blocknum", C.c_int), ("ref_warp_number", C.c_int), ("ref_rate_fill0", C.c_double)]

    def as_dict(self):
        return {n: getattr(self, n) for n, _ in self._fields_}


class MgInfo(C.Structure):
    _fields_ = [(n, C.c_int) for n in "precision n_gpus rank rowA colA row_begin row_end stride".split()] + [
        ("nnz_own", C.c_longlong), ("nnz_other", C.c_longlong), ("overlap", C.c_int), ("has_comm", C.c_int), ("square", C.c_int), ("stream_memops", C.c_int), ("fused_step", C.c_int), ("exchange", C.c_int)]

    def as_dict(self):
        return {n: getattr(self, n) for n, _ in self._fields_}


def build(force=False):
    """Compile libdasp_amd.so for gfx950 with hipcc (cross-compiles without a GPU)."""
    src = os.path.join(_HERE, "csrc")
    if force:
        subprocess.check_call(["make", "-C", src, "-s", "clean"])
    subprocess.check_call(["make", "-C", src, "-s", "-j8", "all"])
    return SO_PATH


_lib = None


def _preload_hip_runtime():
    """One HIP runtime per process.  PyTorch-ROCm wheels bundle their own libamdhip64.so (same
    soname as /opt/rocm's).  If libdasp_amd.so pulled in the system copy first and torch its own
    later, the process would hold two ROCr instances and the second one sees no device.  So when a
    torch wheel is installed, its runtime is mapped first (by path, RTLD_GLOBAL): libdasp_amd.so's
    NEEDED libamdhip64.so.7 then binds to it by soname, and a later `import torch` finds the very
    same file already loaded.  Without torch the system runtime is used."""
    import importlib.util
    import sys
    if "torch" in sys.modules:
        return
    try:
        spec = importlib.util.find_spec("torch")
    except (ImportError, ValueError):
        spec = None
    if spec is None or not spec.origin:
        return
    cand = os.path.join(os.path.dirname(spec.origin), "lib", "libamdhip64.so")
    if os.path.exists(cand):
        try:
            C.CDLL(cand, mode=C.RTLD_GLOBAL)
        except OSError:
            pass


def lib():
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(SO_PATH):
        raise ImportError(
            "dasp_amd: %s is missing. Build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "(hipcc --offload-arch=gfx950). There is no fallback implementation." % SO_PATH)
    _preload_hip_runtime()
    L = C.CDLL(SO_PATH)
    ip, vp = C.POINTER(C.c_int), C.c_void_p
    L.dasp_last_error.restype = C.c_char_p
    L.dasp_version.restype = C.c_char_p
    L.dasp_free.argtypes = [vp]
    L.dasp_options_default.argtypes = [C.POINTER(Options)]
    L.dasp_mmio_allinone_f64.argtypes = [ip, ip, ip, ip, C.POINTER(ip), C.POINTER(ip), C.POINTER(C.POINTER(C.c_double)), C.c_char_p]
    L.dasp_mmio_allinone_f16.argtypes = [ip, ip, ip, ip, C.POINTER(ip), C.POINTER(ip), C.POINTER(C.POINTER(C.c_uint16)), C.c_char_p]
    L.dasp_csr_save.argtypes = [C.c_char_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, vp, vp, vp]
    L.dasp_csr_load.argtypes = [C.c_char_p, C.c_int, ip, ip, ip, ip, C.POINTER(ip), C.POINTER(ip), C.POINTER(vp)]
    L.dasp_plan_create.argtypes = [C.POINTER(vp), C.c_int, C.c_int, C.c_int, C.c_int, vp, vp, vp, C.POINTER(Options)]
    L.dasp_plan_create_device.argtypes = [C.POINTER(vp), C.c_int, C.c_int, C.c_int, C.c_int, vp, vp, vp, C.POINTER(Options)]
    L.dasp_plan_download_array.argtypes = [vp, C.c_char_p, vp, C.c_size_t]
    L.dasp_plan_destroy.argtypes = [vp]
    L.dasp_plan_save.argtypes = [vp, C.c_char_p]
    L.dasp_plan_load.argtypes = [C.POINTER(vp), C.c_char_p]
    L.dasp_plan_order.argtypes = [vp]
    L.dasp_plan_order.restype = ip
    L.dasp_plan_stats.argtypes = [vp, C.POINTER(Stats)]
    L.dasp_plan_y_order.argtypes = [vp]
    L.dasp_plan_x_len.argtypes = [vp]
    L.dasp_plan_x_len.restype = C.c_longlong
    L.dasp_plan_host_array.argtypes = [vp, C.c_char_p, C.POINTER(vp), ip]
    L.dasp_plan_host_array.restype = C.c_longlong
    L.dasp_plan_panel_count.argtypes = [vp]
    L.dasp_plan_panel.argtypes = [vp, C.c_int]
    L.dasp_plan_panel.restype = vp
    L.dasp_plan_panel_range.argtypes = [vp, C.c_int, ip, ip]
    L.dasp_plan_upload.argtypes = [vp]
    L.dasp_plan_drop_host.argtypes = [vp]
    L.dasp_plan_set_stream_policy.argtypes = [vp, C.c_int]
    L.dasp_plan_spmv.argtypes = [vp, vp, vp, vp]
    L.dasp_plan_spmv_acc.argtypes = [vp, vp, vp, vp]
    L.dasp_plan_time.argtypes = [vp, vp, vp, vp, C.c_int, C.c_int, C.POINTER(C.c_double), C.POINTER(C.c_double)]
    L.dasp_plan_time_each.argtypes = [vp, vp, vp, vp, C.c_int, C.c_int, vp]
    L.dasp_plan_time_graph.argtypes = [vp, vp, vp, vp, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_double), C.POINTER(C.c_double)]
    L.dasp_spmv_all_f64.argtypes = [C.c_char_p, vp, vp, vp, vp, vp, vp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_double, C.c_int]
    L.dasp_spmv_all_f16.argtypes = L.dasp_spmv_all_f64.argtypes
    L.dasp_partition_rows.argtypes = [C.c_int, vp, C.c_int, vp]
    L.dasp_selftest_mfma.argtypes = []
    L.dasp_plan_tune_placement.argtypes = [vp, C.c_int, vp, vp, C.POINTER(C.c_double), C.POINTER(C.c_double)]
    L.dasp_mg_unique_id.argtypes = [vp]
    L.dasp_mg_plan_create.argtypes = [C.POINTER(vp), C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, vp, vp, vp, vp, C.POINTER(Options), C.c_int]
    L.dasp_mg_destroy.argtypes = [vp]
    L.dasp_mg_upload.argtypes = [vp]
    L.dasp_mg_comm_init.argtypes = [vp, vp]
    L.dasp_mg_set_x.argtypes = [vp, vp]
    L.dasp_mg_spmv.argtypes = [vp, vp]
    L.dasp_mg_product.argtypes = [vp, vp]
    L.dasp_mg_allgather.argtypes = [vp, vp]
    L.dasp_mg_wait.argtypes = [vp, vp]
    L.dasp_mg_get_y.argtypes = [vp, vp]
    L.dasp_mg_get_y_local.argtypes = [vp, vp]
    for f in (L.dasp_mg_y_local, L.dasp_mg_gathered, L.dasp_mg_x):
        f.argtypes = [vp]
        f.restype = vp
    L.dasp_mg_subplan.argtypes = [vp, C.c_int]
    L.dasp_mg_subplan.restype = vp
    L.dasp_mg_info.argtypes = [vp, C.POINTER(MgInfo)]
    L.dasp_mg_check.argtypes = [vp]
    L.dasp_mg_set_fused.argtypes = [vp, C.c_int]
    L.dasp_mg_set_fake_exchange.argtypes = [vp, C.c_int, C.c_int, vp]
    L.dasp_mg_push_export.argtypes = [vp, vp]
    L.dasp_mg_push_connect.argtypes = [vp, vp]
    L.dasp_mg_set_exchange.argtypes = [vp, C.c_int]
    L.dasp_mg_push_loopback.argtypes = [vp]
    L.dasp_mg_reserved_stream.argtypes = [vp, C.c_int]
    L.dasp_mg_reserved_stream.restype = vp
    L.dasp_synth_dims.argtypes = [C.c_char_p, C.c_double, ip, ip]
    L.dasp_synth_generator.argtypes = [C.c_char_p]
    L.dasp_synth_generator.restype = C.c_char_p
    L.dasp_synth_row_lengths.argtypes = [C.c_char_p, C.c_double, C.c_int, C.c_int, vp]
    L.dasp_synth_rows.argtypes = [C.c_char_p, C.c_double, C.c_int, C.c_int, vp, vp]
    _lib = L
    return L


def check(rc):
    if rc != 0:
        raise DaspError(rc, lib().dasp_last_error().decode("utf-8", "replace"))


EXPORTS = (
    "dasp_last_error dasp_version dasp_mmio_allinone_f64 dasp_mmio_allinone_f16 dasp_free dasp_csr_save dasp_csr_load dasp_options_default "
    "dasp_plan_create dasp_plan_create_device dasp_plan_download_array dasp_plan_destroy dasp_plan_save dasp_plan_load dasp_plan_order dasp_plan_stats dasp_plan_y_order dasp_plan_x_len dasp_plan_panel_count dasp_plan_panel dasp_plan_panel_range dasp_plan_host_array dasp_plan_upload dasp_plan_tune_placement "
    "dasp_plan_drop_host dasp_plan_set_stream_policy dasp_plan_spmv dasp_plan_spmv_acc dasp_plan_time dasp_plan_time_each dasp_plan_time_graph dasp_spmv_all_f64 dasp_spmv_all_f16 dasp_partition_rows "
    "dasp_selftest_mfma dasp_synth_dims dasp_synth_generator dasp_synth_row_lengths dasp_synth_rows "
    "dasp_mg_unique_id dasp_mg_plan_create dasp_mg_destroy dasp_mg_upload dasp_mg_comm_init dasp_mg_set_x dasp_mg_spmv dasp_mg_product dasp_mg_allgather "
    "dasp_mg_wait dasp_mg_get_y dasp_mg_get_y_local dasp_mg_y_local dasp_mg_gathered dasp_mg_x dasp_mg_subplan dasp_mg_info dasp_mg_check dasp_mg_set_fused dasp_mg_set_fake_exchange "
    "dasp_mg_push_export dasp_mg_push_connect dasp_mg_set_exchange dasp_mg_push_loopback dasp_mg_reserved_stream").split()

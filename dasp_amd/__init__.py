"""dasp_amd -- MI355X (gfx950) implementation of the DASP SpMV hot path.

Host preprocessing (C++) and hand-written HIP kernels live in libdasp_amd.so behind the C ABI of
include/dasp_amd.h; this package is the thin Python mirror of the reference's interface.
"""
from ._lib import DaspError, SO_PATH, build  # noqa: F401
from .api import (Plan, csr_load, csr_save, Y_NATURAL, Y_PERMUTED, SYNTH_NAMES, mmio_allinone, partition_rows, selftest_mfma, spmv_all,  # noqa: F401
                  synth_csr, synth_dims, synth_generator, synth_row_lengths)
from . import multi  # noqa: F401,E402

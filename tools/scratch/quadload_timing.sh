# tools/scratch/quadload_timing.sh -- r5, TIMING ONLY: apply tools/proto/quadload_timing.patch, `bash tools/build_variant.sh quadload "-DDASP_EXP_QUADLOAD"`, revert the patch, then run this (the variant's results are wrong by construction)
export SWEEP_ONLY="short rows" SWEEP_PREC=64
echo "== product"; python3 tools/category_sweep.py 2>&1 | grep " us "
echo "== quad loads (timing only; results wrong)"; DASP_AMD_SO=dasp_amd/variants/quadload/libdasp_amd.so python3 tools/category_sweep.py 2>&1 | grep " us "

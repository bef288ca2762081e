export SWEEP_ONLY="short rows" SWEEP_PREC=64
echo "== product"; python3 tools/category_sweep.py 2>&1 | grep " us "
echo "== quad loads (timing only; results wrong)"; DASP_AMD_SO=dasp_amd/variants/quadload/libdasp_amd.so python3 tools/category_sweep.py 2>&1 | grep " us "

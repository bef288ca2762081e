# tools/r5_occ_small_blocks.sh -- r5: are the short medium blocks (rows of 17 / 40) bound by bytes in flight?  The product against builds held to 7 / 8 waves per SIMD
export SWEEP_ONLY="medium rows only,circuit"
for v in "" w7 w8; do echo "== ${v:-product}"; env ${v:+DASP_AMD_SO=dasp_amd/variants/$v/libdasp_amd.so} python3 tools/category_sweep.py 2>&1 | grep " us "; done

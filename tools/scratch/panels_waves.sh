# tools/scratch/panels_waves.sh -- r5: the column-panel kernels (powerlaw_1M f64: 3 panels, 78 registers = 6 waves per SIMD) on builds held to 7 / 8 waves
export DASP_PLACEMENT_TRIALS=4
for w in "powerlaw_1M 64" "powerlaw_1M 64 0.3" "rmat_2M 64" "HV15R-unstructured 64"; do
  for v in "" w7 w8; do env ${v:+DASP_AMD_SO=dasp_amd/variants/$v/libdasp_amd.so} python3 tools/plan_time.py $w 2>&1 | tail -1; done
done

# tools/r5_occupancy_probe2.sh -- r5: as r5_occupancy_probe.sh with LDS sizes that give exactly 6 / 5 / 4 workgroups per CU
for w in "HV15R 1 64 200 20" "nlpkkt160 1 64 200 20"; do
  tools/ab_env.sh "$w" "LD_PRELOAD=dasp_amd/variants/occ/libdasp_amd.so DASP_OCC_LDS=0" "LD_PRELOAD=dasp_amd/variants/occ/libdasp_amd.so DASP_OCC_LDS=26000" "LD_PRELOAD=dasp_amd/variants/occ/libdasp_amd.so DASP_OCC_LDS=30000" "LD_PRELOAD=dasp_amd/variants/occ/libdasp_amd.so DASP_OCC_LDS=36000"
done

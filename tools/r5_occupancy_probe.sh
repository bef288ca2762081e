# tools/r5_occupancy_probe.sh -- r5: the HBM-bound f64 kernels held to fewer workgroups per CU by unused dynamic LDS (a variant build reads DASP_OCC_LDS): how much a lost wave per SIMD costs (profiles/r05_id_encoding.md)
for w in "HV15R 1 64 200 20" "nlpkkt160 1 64 200 20" "Queen_4147 1 64 200 20"; do
  tools/ab_env.sh "$w" "LD_PRELOAD=dasp_amd/variants/occ/libdasp_amd.so DASP_OCC_LDS=0" "LD_PRELOAD=dasp_amd/variants/occ/libdasp_amd.so DASP_OCC_LDS=32768" "LD_PRELOAD=dasp_amd/variants/occ/libdasp_amd.so DASP_OCC_LDS=40960"
done

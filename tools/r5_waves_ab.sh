# tools/r5_waves_ab.sh -- the plain kernels held to 7 / 8 waves per SIMD (72 / 64 VGPRs, a little scratch) against the product (6-7 waves, no scratch)
for w in "HV15R 1 64 200 20" "nlpkkt160 1 64 200 20" "Queen_4147 1 64 200 20" "HV15R-unstructured 1 64 200 20" "powerlaw_1M 1 64 100 10" "rmat_2M 1 16 500 50" "webbase-1M 1 16 2000 100"; do
  tools/ab_env.sh "$w" "X=product" "LD_PRELOAD=dasp_amd/variants/w7/libdasp_amd.so" "LD_PRELOAD=dasp_amd/variants/w8/libdasp_amd.so"
done

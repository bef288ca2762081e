# tools/r5_argptr_ab.sh -- r5: kernel arguments by pointer (the working tree) against the by-value build (dasp_amd/variants/base, built from the previous revision with tools/build_variant.sh), interleaved on one device
for w in "cop20k_A 1 64 3000 200" "webbase-1M 1 16 3000 200" "webbase-1M 1 64 3000 200" "HV15R 1 64 300 30" "nlpkkt160 1 64 300 30" "rmat_2M 1 16 1000 50"; do
  tools/ab_env.sh "$w" "X=new" "LD_PRELOAD=dasp_amd/variants/base/libdasp_amd.so"
done

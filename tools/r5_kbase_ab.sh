# tools/r5_kbase_ab.sh -- r5: one-byte ids above a base per tile column (plan.hpp med_kb) against the build before it (tools/ab/libdasp_amd_base.so, built from the commit before)
# (tools/ab/libdasp_amd_base*.so = the library of the commit before the change under test: `bash tools/build_rev.sh <rev> base && mkdir -p tools/ab && cp dasp_amd/variants/base/libdasp_amd.so tools/ab/<name the script uses>`; git-ignored, removed after the run)
# and against the same build with r4's narrow rule (cid8=2), interleaved on one box
export DASP_PLACEMENT_TRIALS=${TRIALS:-4}       # the fastest of four placements of the arena in every run: the +-8 % lottery out of the comparison
for w in ${WORKLOADS:-nlpkkt160 HV15R Queen_4147 HV15R-unstructured}; do
  for rep in 1 2; do
    DASP_AMD_SO=tools/ab/libdasp_amd_base.so python3 tools/plan_time.py $w 64 2>&1 | tail -1
    python3 tools/plan_time.py $w 64 2>&1 | tail -1
    python3 tools/plan_time.py $w 64 cid8=2 2>&1 | tail -1
  done
done

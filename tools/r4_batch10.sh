#!/bin/bash
mkdir -p gpurun_out/r4
export DASP_AMD_SO=$PWD/dasp_amd/variants/exp/libdasp_amd.so
for hot in 0.7 1.0; do for fm in 0 1; do
echo "== hot_at $hot fence $fm"
DASP_MG_HOT_AT=$hot DASP_MG_STEP2_FENCE=$fm timeout 600 python -m pytest tests -m gpu -x -q -k "one_stream" 2>&1 | tail -2
done; done

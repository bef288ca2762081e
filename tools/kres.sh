#!/bin/bash
# tools/kres.sh -- registers / scratch / occupancy of every dasp_spmv_kernel instantiation (hipcc -Rpass-analysis), one line each
cd "$(dirname "$0")/../dasp_amd/csrc" && /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-parameter -mllvm -amdgpu-mfma-vgpr-form=1 \
  -c kernels.hip -o /tmp/kres.o -Rpass-analysis=kernel-resource-usage 2>&1 | grep -E "error|Function Name|    VGPRs:|ScratchSize|Occupancy" | \
  sed -E 's/.*remark: +//; s/ \[-Rpass.*//' | paste - - - - | grep -E "error|spmv_kernel" | sed -E 's/Function Name: _ZN4dasp16dasp_spmv_kernelI//; s/EEvNS_7DevArgsE//; s/Lb1/1/g; s/Lb0/0/g; s/DF16_/h/; s/^d/d /; s/^h/h /'

#!/bin/bash
# tools/fat_exchange_probe.sh: does the exchange overlap the own-column product?  rank 3 of the 8-way HV15R partition, exchange of 0 / 40 / 60 us
# (tools/mg_step_probe.py), with three stand-ins for the exchange:
#   slim  : a one-wave kernel holding the communication stream (what r2 / early r3 projections used)
#   rccl  : a kernel with the footprint of RCCL's on gfx950 (256 threads, 280 registers, 19.7 KB LDS; DASP_MG_FAKE_CHANNELS workgroups)
#   push  : the direct exchange (mgx.hip) storing into scratch memory of this GPU + a one-wave kernel for the links' share of the time
export PROBE_AG_US=0,40,60
echo "== slim exchange kernel"; timeout 600 python tools/mg_step_probe.py 8 HV15R 3 2>&1 | grep -v "^\[" | tail -7
for ch in 8 32; do echo "== two-launch form, RCCL-footprint exchange kernel, $ch channels"; DASP_MG_FUSED=0 DASP_MG_FAKE_CHANNELS=$ch timeout 600 python tools/mg_step_probe.py 8 HV15R 3 2>&1 | grep -v "^\[" | tail -3; done
echo "== fused step, RCCL-footprint exchange kernel, 8 channels, time-out 50 ms (expected: the in-kernel wait times out)"; DASP_MG_FAKE_CHANNELS=8 DASP_MG_TIMEOUT_MS=50 timeout 600 python tools/mg_step_probe.py 8 HV15R 3 2>&1 | grep -v "^\[" | tail -2 | cut -c1-300
echo "== direct exchange (push), loopback"; PROBE_EXCHANGE=push timeout 600 python tools/mg_step_probe.py 8 HV15R 3 2>&1 | grep -v "^\[" | tail -7
for w in 4 32; do echo "== direct exchange (push), loopback, $w workgroups per destination"; DASP_MG_PUSH_WGS=$w PROBE_EXCHANGE=push timeout 600 python tools/mg_step_probe.py 8 HV15R 3 2>&1 | grep -v "^\[" | tail -6; done

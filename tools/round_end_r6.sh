#!/bin/bash
# tools/round_end_r6.sh -- the measurement batch behind profiles/r06_*: run through gpurun from the repo root, then `ROUND=r06 python tools/round_end_collect.py` here.
#   smoke | full -m gpu suite | PMC traffic of every workload (tools/traffic_all.sh -> profiles/traffic.json at this kernel revision) | bench.py as the driver runs it
#   (compact line + bench_suite.json) | rocprofv3 --kernel-trace --stats of the bench command | MFMA / occupancy counters of four workloads (tools/pmc.sh)
export ROUND=${ROUND:-r06}
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
python -m pytest tests -m gpu -x -q > gpurun_out/round_end_tests.log 2>&1; grep -E "passed|failed" gpurun_out/round_end_tests.log | tail -1
bash tools/traffic_all.sh > gpurun_out/round_end_traffic.log 2>&1
python tools/traffic_collect.py > gpurun_out/round_end_traffic_collect.log 2>&1 && cp profiles/traffic.json gpurun_out/round_end_traffic.json && cp profiles/${ROUND}_traffic.md gpurun_out/round_end_traffic.md
python3 bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/round_end_bench.json.log 2> gpurun_out/round_end_bench.err
cp gpurun_out/bench_suite.json gpurun_out/round_end_bench_suite.json
export TMPDIR=/tmp
rm -rf gpurun_out/round_end_prof
(cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/round_end_prof -- python3 $GRAFT_REPO_ROOT/bench.py --no-suite --no-vendor --steps 20 > $GRAFT_REPO_ROOT/gpurun_out/round_end_prof.log 2>&1)
cp gpurun_out/bench_suite.json gpurun_out/round_end_prof_suite.json
# MFMA utilisation (north_star): one counter group per pass, the program itself after --
B=dasp_amd/bin/dasp_bench
export PMC_GROUPS="SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVES;SQ_INSTS_VALU_MFMA_MOPS_F16 SQ_INSTS_MFMA SQ_INSTS_VALU SQ_ACTIVE_INST_VALU;SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_INSTS_LDS SQ_INSTS_VMEM_RD;GRBM_GUI_ACTIVE"
bash tools/pmc.sh mfma_hv15r64 -- $B HV15R 1 64 20 3 > /dev/null 2>&1
bash tools/pmc.sh mfma_nlp64 -- $B nlpkkt160 1 64 20 3 > /dev/null 2>&1
bash tools/pmc.sh mfma_lj16_dasp -- $B ljournal-2008 1 16 20 3 0.75 0 0 0 0 0 0 0 0 0 0 -1 > /dev/null 2>&1
bash tools/pmc.sh mfma_rmat16 -- $B rmat_2M 1 16 50 3 0.75 0 0 0 0 0 0 0 0 0 0 -1 > /dev/null 2>&1      # (the DASP form: the product's choice for rmat_2M is two-phase since r5)
bash tools/pmc.sh mfma_wb16 -- $B webbase-1M 1 16 200 3 > /dev/null 2>&1
DASP_PMC_KERNELS="dasp_tp_" bash tools/pmc.sh tp_lj16 -- $B ljournal-2008 1 16 20 3 > /dev/null 2>&1
ls gpurun_out/pmc_*.txt
# r6: the two small BASELINE matrices' L2 / L1 counters at the final kernels
export PMC_GROUPS="TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum;TCP_TCC_READ_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_PENDING_STALL_CYCLES_sum;TA_TOTAL_WAVEFRONTS_sum TA_BUSY_avr;GRBM_GUI_ACTIVE"
bash tools/pmc.sh r6_cop64 -- $B cop20k_A 1 64 200 20 > /dev/null 2>&1
bash tools/pmc.sh r6_wb16 -- $B webbase-1M 1 16 200 20 > /dev/null 2>&1
ls gpurun_out/pmc_r6_*.txt

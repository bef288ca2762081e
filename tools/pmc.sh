#!/bin/bash
# tools/pmc.sh <tag> -- <program> [args]   : several rocprofv3 --pmc passes (one counter group per run), summary to gpurun_out/pmc_<tag>.txt
tag=$1; shift; shift
root=${GRAFT_REPO_ROOT:-$PWD}; out=$root/gpurun_out/pmc_$tag; mkdir -p "$out"; export TMPDIR=/tmp; cd "$root"
groups=(
 "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM_RD SQ_INSTS_VALU"
 "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum"
 "TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_PENDING_STALL_CYCLES_sum"
 "TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum"
 "TA_TOTAL_WAVEFRONTS_sum TA_BUSY_avr"
 "GRBM_GUI_ACTIVE"
 "TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_TRANSLATION_HIT_sum TCP_READ_TAGCONFLICT_STALL_CYCLES_sum TCP_TCR_TCP_STALL_CYCLES_sum"
 "SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_VALU_MFMA_BUSY_CYCLES SQ_INST_LEVEL_VMEM SQ_INSTS_SMEM SQ_INSTS_SALU SQ_IFETCH"
)
# PMC_GROUPS="A B;C D" replaces the default groups
if [ -n "$PMC_GROUPS" ]; then IFS=';' read -r -a groups <<< "$PMC_GROUPS"; fi
i=0
for g in "${groups[@]}"; do
  # a group the hardware cannot schedule makes rocprofv3 abort and then hang in its signal handler: always bound it
  timeout -k 5 180 rocprofv3 --kernel-trace --pmc $g --output-format csv -d "$out/g$i" -- "$@" > "$out/g$i.log" 2>&1
  i=$((i+1))
done
python3 - "$out" <<'PY' > "$root/gpurun_out/pmc_$tag.txt"
import csv,glob,sys,collections,os
out=sys.argv[1]
agg=collections.defaultdict(list); dur=[]
for f in glob.glob(out+"/g*/*/*counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        if any(k in r["Kernel_Name"] for k in os.environ.get("DASP_PMC_KERNELS", "dasp_spmv").split(",")):
            agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
            dur.append(int(r["End_Timestamp"])-int(r["Start_Timestamp"]))
print("dispatch duration under PMC: median %.1f us over %d" % (sorted(dur)[len(dur)//2]/1e3, len(dur)))
for k in sorted(agg): v=agg[k]; print("%-44s mean %.6g  (n=%d)" % (k, sum(v)/len(v), len(v)))
PY
cat "$root/gpurun_out/pmc_$tag.txt"

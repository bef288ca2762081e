# tools/mg_24way.sh -- the 2-way and 4-way partitions of BASELINE config 5 through the same one-GPU emulation as the 8-way tables (two-plan fused step, direct exchange in loopback)
mkdir -p gpurun_out/r4
for w in HV15R Queen_4147; do for n in 2 4; do
  PROBE_FULL=1 PROBE_EXCHANGE=push PROBE_AG_US=0,30,60 timeout 1200 python tools/mg_step_probe.py $n $w all > gpurun_out/r4/mg_${n}way_$w.log 2>&1
  echo "== $w $n-way"; grep -E "1-GPU step|max over ranks fused" gpurun_out/r4/mg_${n}way_$w.log | cut -c1-100
done; done

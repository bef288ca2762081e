# tools/ready_event_ab.sh -- the fused two-plan step with "y ready" published by a one-lane kernel behind the step kernel (default) against the last workgroup
# arriving at a counter inside it (DASP_MG_READY_KERNEL=0); rank 1 of the 8-way partitions, direct exchange in loopback
for w in HV15R Queen_4147; do
for rk in 0 1 0 1; do
  echo "== $w DASP_MG_READY_KERNEL=$rk"
  DASP_MG_READY_KERNEL=$rk PROBE_FULL=1 PROBE_EXCHANGE=push PROBE_AG_US=0,30,45 timeout 600 python tools/mg_step_probe.py 8 $w 1 2>&1 | grep -E "fused" | grep "max over" | cut -c1-200
done
done

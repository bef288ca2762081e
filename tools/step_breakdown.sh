# tools/step_breakdown.sh -- where the fused two-plan step's time goes (rank 1 of the 8-way HV15R partition): position of the waiting workgroups in the grid
for at in 0.5 0.65 0.8 0.9 1.0; do
  echo "== DASP_MG_POLL_AT=$at"
  DASP_MG_POLL_AT=$at PROBE_FULL=1 PROBE_EXCHANGE=push PROBE_AG_US=0,30,45 timeout 600 python tools/mg_step_probe.py 8 HV15R 1 2>&1 | grep -E "^rank 1" | sed 's/rows.*fused_ok 1 |//; s/2launch.*| step kernel alone/| step kernel alone/; s/| own alone.*//' | cut -c1-220
done

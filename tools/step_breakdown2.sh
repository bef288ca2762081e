for no in 0 1; do
  echo "== NOOTHER=$no"
  DASP_AMD_SO=dasp_amd/variants/exp/libdasp_amd.so DASP_MG_STEP_NOOTHER=$no PROBE_FULL=1 PROBE_EXCHANGE=push PROBE_AG_US=0 timeout 600 python tools/mg_step_probe.py 2 Queen_4147 1 2>&1 | grep -E "^rank 1" | sed 's/.*| step kernel alone/step kernel alone/' | cut -c1-260
done

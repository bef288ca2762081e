#!/bin/bash
# bench.py --gpus 8 with all eight ranks sharing this box's one GPU (test hook): the whole multi-rank flow -- direct exchange over hipIpc, first-contact checks,
# chained and random-x verification -- in both overlap modes
mkdir -p gpurun_out/r4
for ov in 1 2; do
DASP_BENCH_OVERLAP=$ov DASP_BENCH_SHARE_GPU=1 timeout 900 python bench.py --gpus 8 --scale 1.0 --steps 30 --warmup 5 > gpurun_out/r4/share8_ov$ov.json.log 2> gpurun_out/r4/share8_ov$ov.err
echo "== overlap $ov rc=$?"; tail -c 400 gpurun_out/r4/share8_ov$ov.err
python - <<EOF
import json
try:
    d=json.loads(open("gpurun_out/r4/share8_ov$ov.json.log").read().strip().splitlines()[-1])
    print({k:d.get(k) for k in ("value","ms_per_step","verified","error")}, d["config"].get("step_form"), d["config"].get("exchange"), d.get("verified_random_x",{}).get("ok"), d.get("exchange_ms"))
except Exception as e: print("no json", e)
EOF
done

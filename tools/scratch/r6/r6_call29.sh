#!/bin/bash
# r6_call29 -- the driver's round-end sequence on the final tree: GPU tests, smoke(), the default bench line
export PYTHONPATH=$PWD
mkdir -p gpurun_out/r6
timeout 1500 python3 -m pytest tests -x -q -m gpu > gpurun_out/r6/final_gpu_tests.log 2>&1; tail -2 gpurun_out/r6/final_gpu_tests.log
python3 -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
python3 bench.py > gpurun_out/r6/final_bench_line.json 2> gpurun_out/r6/final_bench.err; echo "bench rc $?"; wc -c gpurun_out/r6/final_bench_line.json
python3 - <<'PY'
import json
d = json.loads(open("gpurun_out/r6/final_bench_line.json").read().strip().splitlines()[-1])
print({k: d[k] for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "dtype", "scaling", "vs_baseline")})
print("roofline", {k: d["roofline"][k] for k in ("bound", "achieved", "peak", "unit", "frac", "traffic")})
print("cpu_baseline", d["cpu_baseline"])
PY

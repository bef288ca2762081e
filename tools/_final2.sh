python -m pytest tests -m gpu -x -q > gpurun_out/r3_final_tests2.log 2>&1; tail -2 gpurun_out/r3_final_tests2.log
bash tools/traffic_all.sh > gpurun_out/r3_traffic_all2.log 2>&1; tail -3 gpurun_out/r3_traffic_all2.log | cut -c1-120
python bench.py > gpurun_out/r3_final_bench2.json.log 2> gpurun_out/r3_final_bench2.err; tail -c 200 gpurun_out/r3_final_bench2.json.log; echo

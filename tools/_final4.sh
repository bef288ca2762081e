python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
python -m pytest tests -m gpu -x -q > gpurun_out/r3_final_tests4.log 2>&1; tail -2 gpurun_out/r3_final_tests4.log
bash tools/traffic_all.sh > gpurun_out/r3_traffic_all4.log 2>&1; tail -2 gpurun_out/r3_traffic_all4.log | cut -c1-100

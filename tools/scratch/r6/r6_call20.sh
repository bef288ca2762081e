#!/bin/bash
# r6_call20 -- the hybrid-window rule once more at today's kernels (r2's cases, tools/hybrid_probe.py): default / hybrid forced / no windows
export PYTHONPATH=$PWD
for args in "2000000 14 3000 0.1 64" "2000000 14 3000 0.1 16" "2000000 14 8000 0.03 64" "2000000 14 8000 0.03 16" "500000 14 3000 0.1 64" "500000 14 3000 0.1 16" "200000 30 2000 0.05 64" "4000000 14 3000 0.1 64"; do
  echo "== $args"; sed 's/x_window=163840/x_window=163584/' tools/hybrid_probe.py > /tmp/hp.py; timeout 300 python3 /tmp/hp.py $args 2>&1 | grep -v amdgpu.ids
done

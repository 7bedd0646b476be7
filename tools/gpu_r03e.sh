#!/bin/bash
# round 3, visit e: D10 with the cheaper per-surfel set-up, A/B over thresholds; HIP-graph replay probe
export TMPDIR=/tmp
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "parity or tile_cull or c3 or full_size" > gpurun_out/r03e_pytest.log 2>&1
echo "pytest rc=$?" >> gpurun_out/r03e_pytest.log
tail -3 gpurun_out/r03e_pytest.log
for rep in 1 2; do
  for v in 1 3 6; do
    SLS_TILE_CULL_MIN=$v timeout 200 python bench.py --no-cpu-baseline --no-extras 2>/dev/null | python -c "
import json, sys
d = json.loads(sys.stdin.read())
print('tile_cull_min=$v', d['value'], d['config']['ms_per_iteration'], d['config']['R'], d['config']['R_eff'], {k: v['avg_us'] for k, v in d['kernels'].items()})"
  done
done
timeout 120 python tools/graph_probe.py 50000 64 1024 400 2>&1 | grep -v "^$" | tail -5
timeout 120 python tools/graph_probe.py 170000 64 1024 400 2>&1 | grep -v "^$" | tail -5
timeout 120 python tools/graph_probe.py 500000 64 2048 200 2>&1 | grep -v "^$" | tail -5

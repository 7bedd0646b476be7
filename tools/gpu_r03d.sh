#!/bin/bash
# round 3, visit d: D10 thresholds A/B (cheaper owner search, packed emit record), enqueue rate at the reference's sizes
export TMPDIR=/tmp
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_timed_path.py -m gpu -q -x > gpurun_out/r03d_pytest.log 2>&1
echo "pytest rc=$?" >> gpurun_out/r03d_pytest.log
tail -3 gpurun_out/r03d_pytest.log
for rep in 1 2; do
  for v in 1 3 6 9; do
    SLS_TILE_CULL_MIN=$v timeout 200 python bench.py --no-cpu-baseline --no-extras 2>/dev/null | python -c "
import json, sys
d = json.loads(sys.stdin.read())
print('tile_cull_min=$v', d['value'], d['config']['ms_per_iteration'], d['config']['R'], d['config']['R_eff'], {k: v['avg_us'] for k, v in d['kernels'].items()})"
  done
done
for v in 1 6; do
SLS_TILE_CULL_MIN=$v timeout 200 python bench.py --keyframes 1 --no-cpu-baseline --no-extras 2>/dev/null | python -c "
import json, sys
d = json.loads(sys.stdin.read())
print('single keyframe tile_cull_min=$v', d['value'], d['config']['ms_per_iteration'], d['config']['R'], d['config']['R_eff'], {k: v['avg_us'] for k, v in d['kernels'].items()})"
done
timeout 120 python tools/enqueue_rate.py 50000 64 1024 400 2>&1 | grep -v "^$" | head -30
timeout 120 python tools/enqueue_rate.py 500000 64 2048 200 2>&1 | grep "us/iter"

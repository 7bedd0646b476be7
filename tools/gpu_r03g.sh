#!/bin/bash
# round 3, visit g: merged preprocess + window-sort launch (A/B), sparse exchange over gathered bitmaps, full suite
export TMPDIR=/tmp
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
timeout 900 python -m pytest tests -m gpu -q > gpurun_out/r03g_pytest.log 2>&1
echo "pytest rc=$?" >> gpurun_out/r03g_pytest.log
grep -E "passed|failed|FAILED|Error" gpurun_out/r03g_pytest.log | tail -12
for rep in 1 2; do
  for v in 0 1; do
    SLS_NO_MERGED_SORT=$v timeout 200 python bench.py --no-cpu-baseline --no-extras 2>/dev/null | python -c "
import json, sys
d = json.loads(sys.stdin.read())
print('no_merged_sort=$v', d['value'], d['config']['ms_per_iteration'], {k: v['avg_us'] for k, v in d['kernels'].items()})"
  done
done
for cfg in "50000 64 1024" "170000 64 1024"; do
  set -- $cfg
  for v in 0 1; do
  SLS_NO_MERGED_SORT=$v timeout 200 python bench.py --n $1 --height $2 --width $3 --keyframes 1 --no-cpu-baseline --no-extras 2>/dev/null | python -c "
import json, sys
d = json.loads(sys.stdin.read())
print('$1 $2x$3 single keyframe no_merged_sort=$v', d['value'], d['config']['ms_per_iteration'], {k: v['avg_us'] for k, v in d['kernels'].items()})"
  done
done
for m in allreduce sparse; do
SLS_BENCH_BACKEND=gloo timeout 300 python bench.py --gpus 2 --dp-mode $m --steps 5 --warmup 2 --no-cpu-baseline --no-extras 2> gpurun_out/r03g_dp_$m.err | python -c "
import json, sys
try:
    d = json.loads(sys.stdin.read())
    print('$m', d['value'], d['config']['ms_per_iteration'], d['comm'], d['config']['repeated_iterations'])
except Exception as e:
    print('$m failed', e)"
done

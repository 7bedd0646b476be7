#!/bin/bash
# round 3, visit f: the whole -m gpu suite (sparse exchange, D10 default 3), 2-rank gloo bench of the three exchange schemes
export TMPDIR=/tmp
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
timeout 900 python -m pytest tests -m gpu -q > gpurun_out/r03f_pytest.log 2>&1
echo "pytest rc=$?" >> gpurun_out/r03f_pytest.log
grep -E "passed|failed|FAILED|Error" gpurun_out/r03f_pytest.log | tail -12
for m in allreduce rs_ag sparse; do
SLS_BENCH_BACKEND=gloo timeout 300 python bench.py --gpus 2 --dp-mode $m --steps 5 --warmup 2 --no-cpu-baseline --no-extras 2> gpurun_out/r03f_dp_$m.err | python -c "
import json, sys
try:
    d = json.loads(sys.stdin.read())
    print('$m', d['value'], d['config']['ms_per_iteration'], d['comm'], d['config']['repeated_iterations'])
except Exception as e:
    print('$m failed', e)"
tail -3 gpurun_out/r03f_dp_$m.err
done

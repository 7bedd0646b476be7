#!/bin/bash
# One GPU-box visit of round 4 (edited per visit; the generic pieces are tools/gpu_round.sh, profile_round.sh, ab_*.sh).
TAG=${1:-r04x}
export TMPDIR=/tmp
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
bash tools/profile_round.sh $TAG > gpurun_out/${TAG}_profile_round.log 2>&1
tail -25 gpurun_out/${TAG}_profile_round.log
for m in 1 2; do
  SLS_DETERMINISTIC=$m timeout 120 python bench.py --no-cpu-baseline --no-extras 2>/dev/null | python -c "
import json, sys
d = json.loads(sys.stdin.read())
print('SLS_DETERMINISTIC=$m', d['value'], d['config']['ms_per_iteration'], {k: v['avg_us'] for k, v in d['kernels'].items()}, d['config']['repeated_iterations'])"
done
timeout 300 python -m pytest tests -m gpu -q -x -k "deterministic or lagged or repair" 2>&1 | tail -2

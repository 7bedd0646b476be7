#!/bin/bash
# One GPU-box visit of round 4 (edited per visit; the generic pieces are tools/gpu_round.sh, profile_round.sh, ab_*.sh).
TAG=${1:-r04x}
export TMPDIR=/tmp
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
timeout 200 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
for extra in "--dp-mode sparse" "--dp-mode sparse --dp-overlap" "" ; do
  echo "== bench --gpus 2 (gloo on one GPU) $extra"
  SLS_BENCH_BACKEND=gloo timeout 300 python bench.py --gpus 2 --steps 5 --warmup 2 $extra 2>/tmp/err.txt | python -c "
import json, sys
d = json.loads(sys.stdin.read())
print(d['value'], d['config']['ms_per_iteration'], d['dp_mode'], d.get('dp_overlap'), d['dp_calibration_ms'], d['comm'])" || tail -5 /tmp/err.txt
done
timeout 600 python -m pytest tests -m gpu -q 2>&1 | tail -3

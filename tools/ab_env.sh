#!/bin/bash
# A/B of environment switches inside ONE gpurun call, at three sizes:   VARIANTS="_ SLS_NO_DIRECT_BIN=1" bash tools/ab_env.sh
# ("_" = no switch).  Each command under its own timeout; REPS interleaved repetitions (default 2); prints Msplats/s,
# ms per step and the kernels whose name contains $KERNELS (default: all).
export TMPDIR=/tmp
cd "$(dirname "$0")/.."
for shape in ${SHAPES:-"500000,64,2048" "170000,64,1024" "50000,64,1024"}; do
  IFS=, read n h w <<< "$shape"
  for rep in $(seq ${REPS:-2}); do
    for v in $VARIANTS; do
      envs=""; [ "$v" != "_" ] && envs="${v//+/ }"
      env $envs timeout 90 python bench.py --no-cpu-baseline --no-extras --n $n --height $h --width $w 2>/dev/null | python -c "
import json, os, sys
d = json.loads(sys.stdin.read()); sel = os.environ.get('KERNELS', '')
print('$n ${h}x$w [$v]', d['value'], d['config']['ms_per_iteration'], {k: v['avg_us'] for k, v in d['kernels'].items() if sel in k}, d['config']['repeated_iterations'])" || echo "$n ${h}x$w [$v] FAILED"
    done
  done
done

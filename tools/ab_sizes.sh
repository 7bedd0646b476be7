#!/bin/bash
# A/B of library builds (gpurun_tmp_<name>.so, as tools/ab_bench.sh) at three sizes: VARIANTS="a b" bash tools/ab_sizes.sh
# Every command runs under its own short timeout: a hung variant costs a minute, not the visit.
export TMPDIR=/tmp
cd "$(dirname "$0")/.."
for shape in "500000 64 2048" "170000 64 1024" "50000 64 1024"; do
  set -- $shape
  for rep in 1 2; do
    for v in $VARIANTS; do
      cp gpurun_tmp_$v.so splat_loam_amd/libsls_hip.so
      timeout 60 python bench.py --no-cpu-baseline --no-extras --n $1 --height $2 --width $3 2>/dev/null | python -c "
import json, sys
d = json.loads(sys.stdin.read())
print('$1 $2x$3 $v', d['value'], d['ms_per_step'], {k: v['avg_us'] for k, v in d['kernels'].items() if 'render' in k})" || echo "$1 $2x$3 $v FAILED"
    done
  done
done

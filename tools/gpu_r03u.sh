#!/bin/bash
# visit u: dense forward rounds from block masks — parity, then A/B against the three-launch pass at three sizes
export TMPDIR=/tmp
cd "$(dirname "$0")/.."
timeout 900 python -m pytest tests -m gpu -q 2>&1 | tail -3; SLS_BLOCK_MASKS=1 timeout 900 python -m pytest tests -m gpu -q 2>&1 | tail -3
ab() {  # n h w
  for rep in 1 2; do
    for v in 1 0; do
      SLS_BLOCK_MASKS=$((v + 1)) timeout 200 python bench.py --no-cpu-baseline --no-extras --n $1 --height $2 --width $3 2>/dev/null | python -c "
import json, sys
d = json.loads(sys.stdin.read())
print('$1 $2x$3 block_masks=$((v + 1)) (1 always, 2 never)', d['value'], d['ms_per_step'], {k: v['avg_us'] for k, v in d['kernels'].items() if 'sort_sc' in k or 'emit' in k or 'render' in k})"
    done
  done
}
ab 500000 64 2048
ab 170000 64 1024
ab 50000 64 1024

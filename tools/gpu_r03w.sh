#!/bin/bash
# visit w: does the tile-level footprint test (D10) still pay now that both tile kernels run dense rounds?
export TMPDIR=/tmp
cd "$(dirname "$0")/.."
for shape in "500000 64 2048" "170000 64 1024" "50000 64 1024"; do
  set -- $shape
  for rep in 1 2; do
    for v in 1 3 6; do
      SLS_TILE_CULL_MIN=$v timeout 200 python bench.py --no-cpu-baseline --no-extras --n $1 --height $2 --width $3 2>/dev/null | python -c "
import json, sys
d = json.loads(sys.stdin.read())
print('$1 $2x$3 tile_cull_min=$v', d['value'], d['ms_per_step'], d['config']['R'], {k: v['avg_us'] for k, v in d['kernels'].items() if k in ('preprocess_fwd', 'emit_keys', 'sort_scatter', 'render_fwd', 'render_bwd')})"
    done
  done
done

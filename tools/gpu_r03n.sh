#!/bin/bash
# visit n: fixed cost of every kernel of the iteration (almost no surfels)
export TMPDIR=/tmp
cd "$(dirname "$0")/.."
for shape in "2000 64 2048" "2000 64 1024" "50000 64 1024"; do
  set -- $shape
  timeout 200 python bench.py --no-cpu-baseline --no-extras --n $1 --height $2 --width $3 2>/dev/null | python -c "
import json, sys
d = json.loads(sys.stdin.read())
k = {k: v['avg_us'] for k, v in d['kernels'].items()}
print('$1 $2x$3', d['ms_per_step'] * 100, 'us/iter; kernel sum', round(sum(k.values()), 1), k)"
done

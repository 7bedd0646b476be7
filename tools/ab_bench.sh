#!/bin/bash
# A/B of library builds inside ONE gpurun call (numbers from different boxes differ by 2-3 %, more than most of
# the changes worth measuring).  Build the variants as ./gpurun_tmp_<name>.so (tools/build_variant.sh <name> [make variables];
# they travel with the snapshot, git ignores them), then
#   gpurun -- 'bash tools/ab_bench.sh A B [C ...]'
# Each variant is benched REPS times (default 2), interleaved, at every shape of $SHAPES (default: BASELINE config 3);
# prints Msplats/s, ms per iteration and the kernels whose name contains $KERNELS (default: every kernel).
export TMPDIR=/tmp
cd "$(dirname "$0")/.."
cp splat_loam_amd/libsls_hip.so /tmp/ab_keep.so
for shape in ${SHAPES:-"500000,64,2048"}; do
  IFS=, read n h w <<< "$shape"
  for rep in $(seq ${REPS:-2}); do
    for v in "$@"; do
      cp gpurun_tmp_$v.so splat_loam_amd/libsls_hip.so
      timeout 200 python bench.py --no-cpu-baseline --no-extras --n $n --height $h --width $w 2>/dev/null | python -c "
import json, os, sys
d = json.loads(sys.stdin.read()); sel = os.environ.get('KERNELS', '')
print('$n ${h}x$w [$v]', d['value'], d['config']['ms_per_iteration'], {k: v['avg_us'] for k, v in d['kernels'].items() if sel in k})" || echo "$n [$v] FAILED"
    done
  done
done
cp /tmp/ab_keep.so splat_loam_amd/libsls_hip.so

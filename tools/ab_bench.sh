#!/bin/bash
# A/B of library builds inside ONE gpurun call (numbers from different boxes differ by 2-3 %, more than most of
# the changes worth measuring).  Build the variants as ./gpurun_tmp_<name>.so (make -C splat_loam_amd/csrc
# OUT=$PWD/gpurun_tmp_<name>.so [FAST='$(COMMON) ... -D...']; they travel with the snapshot, git ignores them), then
#   gpurun -- 'bash tools/ab_bench.sh A B [C ...]'
# Each variant is benched REPS times (default 2), interleaved; prints Msplats/s, ms per step and the kernels
# whose name contains $KERNELS (default: every kernel).
export TMPDIR=/tmp
cd "$(dirname "$0")/.."
for rep in $(seq ${REPS:-2}); do
  for v in "$@"; do
    cp gpurun_tmp_$v.so splat_loam_amd/libsls_hip.so
    timeout 200 python bench.py --no-cpu-baseline --no-extras 2>/dev/null | python -c "
import json, os, sys
d = json.loads(sys.stdin.read()); sel = os.environ.get('KERNELS', '')
print('$v', d['value'], d['ms_per_step'], {k: v['avg_us'] for k, v in d['kernels'].items() if sel in k})"
  done
done

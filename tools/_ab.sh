export TMPDIR=/tmp
cd /root/repo
for rep in 1 2; do
for v in "$@"; do
cp gpurun_tmp_$v.so splat_loam_amd/libsls_hip.so
timeout 200 python bench.py --no-cpu-baseline --no-extras 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('$v', d['value'], d['ms_per_step'], {k: v['avg_us'] for k,v in d['kernels'].items() if 'sort' in k})"
done
done

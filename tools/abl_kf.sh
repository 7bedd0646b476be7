export TMPDIR=/tmp
cp splat_loam_amd/libsls_hip.so /tmp/ab_keep.so
for kf in 0 5 6 7; do for v in abl1 abl2; do
cp gpurun_tmp_$v.so splat_loam_amd/libsls_hip.so
SLS_BENCH_FIRST_KEYFRAME=$kf timeout 200 python bench.py --no-cpu-baseline --no-extras --keyframes 1 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('kf $kf [$v]', d['config']['ms_per_iteration'], {k: v['avg_us'] for k, v in d['kernels'].items() if k in ('render_fwd','render_bwd','knn','bin_direct','resort')}, d['config']['R'], d['config']['R_eff'])"
done; done
cp /tmp/ab_keep.so splat_loam_amd/libsls_hip.so

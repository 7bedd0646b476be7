#!/bin/bash
TAG=${1:-r06x}
export TMPDIR=/tmp
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
for i in 1 2 3; do
timeout 600 python bench_extras.py dropin 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read())['dropin']
for k,v in d.items():
    if isinstance(v,dict) and 'all_planes' in v:
        print('$i',k,'fwd',v['all_planes']['rasterizer_fwd_ms'],'fwd+bwd',v['all_planes']['rasterizer_fwd_bwd_ms'],'lean',v['lean_allmap']['rasterizer_fwd_bwd_ms'],'staged',v.get('staged_all_planes',{}).get('rasterizer_fwd_bwd_ms'),'hooked',v.get('hooked_mapper'))
    elif k=='pinned_to_four_cores':
        for kk,vv in v.items():
            if isinstance(vv,dict): print('$i pinned',kk,'fwd',vv['all_planes']['rasterizer_fwd_ms'],'fwd+bwd',vv['all_planes']['rasterizer_fwd_bwd_ms'],'lean',vv['lean_allmap']['rasterizer_fwd_bwd_ms'])
    else: print('$i',k,v)
"
done | tee gpurun_out/${TAG}_dropin.txt

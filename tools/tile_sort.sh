#!/bin/bash
# Experiment for DESIGN.md section 9 ("binning without a global depth order"): the bench scenes' tile-list lengths ->
# tools/micro/tile_sort.hip (hipcc --offload-arch=gfx950 -O3 tools/micro/tile_sort.hip -o gpurun_tmp_tile_sort, here).
#   gpurun -- 'bash tools/tile_sort.sh'   ->  gpurun_out/tile_sort.txt
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
mkdir -p gpurun_out
BIN=$PWD/gpurun_tmp_tile_sort
[ -x "$BIN" ] || { echo "build $BIN first"; exit 1; }
for shape in "500000 64 2048" "170000 64 1024" "50000 64 1024"; do
  set -- $shape
  timeout 200 python - $1 $2 $3 > /tmp/lengths_$1.txt <<'PY'
import sys, os
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
import numpy as np, torch
from splat_loam_amd import synth
from helpers import hip_forward
N, H, W = (int(x) for x in sys.argv[1:4])
sc = synth.make_scene(N, H, W, seed=0)
view, proj = synth.camera_matrices(sc["K"], None)
st, t = hip_forward(torch.device("cuda:0"), sc, view, proj, H, W)
r = st.ranges.cpu().numpy().reshape(-1, 2).astype(np.int64)
print("\n".join(str(int(b - a)) for a, b in r))
PY
  echo "== $1 surfels $2x$3: $(wc -l < /tmp/lengths_$1.txt) tiles, lengths min/median/max $(sort -n /tmp/lengths_$1.txt | awk '{a[NR]=$1} END {print a[1] "/" a[int((NR+1)/2)] "/" a[NR]}')"
  timeout 120 "$BIN" < /tmp/lengths_$1.txt
done 2>&1 | grep -v amdgpu.ids | tee gpurun_out/tile_sort.txt

#!/bin/bash
# round 3, visit c: D10 (tile-level cull) parity + A/B, the full -m gpu suite, the reworked bench line, VCC calibration row
export TMPDIR=/tmp
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
timeout 900 python -m pytest tests -m gpu -q -s > gpurun_out/r03c_pytest.log 2>&1
echo "pytest rc=$?" >> gpurun_out/r03c_pytest.log
grep -E "pix_offset|config 4|passed|failed|FAILED|Error" gpurun_out/r03c_pytest.log | tail -20
for rep in 1 2; do
  for v in 0 1; do
    SLS_NO_TILE_CULL=$v timeout 200 python bench.py --no-cpu-baseline --no-extras 2>/dev/null | python -c "
import json, sys
d = json.loads(sys.stdin.read())
print('no_tile_cull=$v', d['value'], d['config']['ms_per_iteration'], d['config']['R'], d['config']['R_eff'], {k: v['avg_us'] for k, v in d['kernels'].items()})"
  done
done
for cfg in "50000 64 1024" "170000 64 1024"; do
  set -- $cfg
  timeout 200 python bench.py --n $1 --height $2 --width $3 --no-cpu-baseline --no-extras 2>/dev/null | python -c "
import json, sys
d = json.loads(sys.stdin.read())
print('$1 $2x$3', d['value'], d['config']['ms_per_iteration'], {k: v['avg_us'] for k, v in d['kernels'].items()})"
done
timeout 600 python bench.py > gpurun_out/r03c_bench.json 2> gpurun_out/r03c_bench.err
tail -25 gpurun_out/r03c_bench.err
python - <<'PY'
import json
try:
    d = json.load(open("gpurun_out/r03c_bench.json"))
    print(d["value"], d["ms_per_step"], d["config"]["ms_per_iteration"])
    print({k: v for k, v in d["roofline"].items() if k not in ("note", "valu")})
    print(d["extras"]); print(d["cpu_baseline"])
except Exception as e:
    print("bench parse failed", e)
PY
bash tools/valu_calib.sh > gpurun_out/r03c_valu.log 2>&1
grep -E "cndmask" gpurun_out/r03c_valu.log | tail -12

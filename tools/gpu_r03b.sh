#!/bin/bash
# round 3, visit b: pix_offset / RCCL / rule-based SLAM tests, the reworked bench line, cndmask calibration rows
export TMPDIR=/tmp
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
timeout 900 python -m pytest tests -m gpu -q -x -s > gpurun_out/r03b_pytest.log 2>&1
echo "pytest rc=$?" >> gpurun_out/r03b_pytest.log
grep -E "pix_offset|config 4|passed|failed|Error|error" gpurun_out/r03b_pytest.log | tail -20
timeout 400 python bench.py > gpurun_out/r03b_bench.json 2> gpurun_out/r03b_bench.err
tail -12 gpurun_out/r03b_bench.err
python - <<'PY'
import json
try:
    d = json.load(open("gpurun_out/r03b_bench.json"))
    print(d["value"], d["ms_per_step"], d["config"]["ms_per_iteration"], {k: v["avg_us"] for k, v in d["kernels"].items()})
    print({k: v for k, v in d["roofline"].items() if k not in ("note", "valu")})
    print(d["extras"]); print(d["cpu_baseline"])
except Exception as e:
    print("bench parse failed", e)
PY
bash tools/valu_calib.sh > gpurun_out/r03b_valu.log 2>&1
grep -E "cndmask" gpurun_out/r03b_valu.log | tail -12

#!/bin/bash
# round 3, visit a: VALU calibration, the -m gpu suite, the bench line, per-kernel breakdown at the reference's sizes
export TMPDIR=/tmp
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
bash tools/valu_calib.sh > gpurun_out/r03a_valu.log 2>&1
tail -70 gpurun_out/r03a_valu.log
timeout 900 python -m pytest tests -m gpu -q -x > gpurun_out/r03a_pytest.log 2>&1
echo "pytest rc=$?" >> gpurun_out/r03a_pytest.log
tail -8 gpurun_out/r03a_pytest.log
timeout 300 python bench.py > gpurun_out/r03a_bench.json 2> gpurun_out/r03a_bench.err
for cfg in "50000 64 1024" "170000 64 1024" "150000 128 1024"; do
  set -- $cfg
  timeout 200 python bench.py --n $1 --height $2 --width $3 --no-cpu-baseline --no-extras --steps 200 --warmup 20 > gpurun_out/r03a_size_$1_$2x$3.json 2> /dev/null
done
python - <<'PY'
import json, glob
for f in sorted(glob.glob("gpurun_out/r03a_bench.json") + glob.glob("gpurun_out/r03a_size_*.json")):
    try:
        d = json.load(open(f))
        print(f, d["value"], d["ms_per_step"], {k: v["avg_us"] for k, v in d["kernels"].items()})
    except Exception as e:
        print(f, "parse failed", e)
PY

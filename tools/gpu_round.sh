#!/bin/bash
# One GPU-box visit: the whole -m gpu suite, the default bench line, optionally a tag for the outputs.
#   gpurun -- 'bash tools/gpu_round.sh r02a [pytest-args]'
TAG=${1:-run}; shift
export TMPDIR=/tmp
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
timeout 900 python -m pytest tests -m gpu -q -x -s "$@" > gpurun_out/${TAG}_pytest.log 2>&1
echo "pytest rc=$?" >> gpurun_out/${TAG}_pytest.log
tail -5 gpurun_out/${TAG}_pytest.log
timeout 600 python bench.py > gpurun_out/${TAG}_bench.json 2> gpurun_out/${TAG}_bench.err
tail -c 400 gpurun_out/${TAG}_bench.err
python - <<PY
import json
try:
    d = json.load(open("gpurun_out/${TAG}_bench.json"))
    print(d["value"], d["ms_per_step"], {k: v["avg_us"] for k, v in d["kernels"].items()})
    print(d["roofline"])
except Exception as e:
    print("bench parse failed", e)
PY

#!/bin/bash
TAG=${1:-r06x}
export TMPDIR=/tmp
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_fused_render.py tests/test_fused_mapper.py "tests/test_gpu_parity.py::test_workspace_path_second_backward_and_held_graph" "tests/test_gpu_parity.py::test_knn_bitexact" -m gpu -q -x 2>&1 | tail -15 | tee gpurun_out/${TAG}_pytest.log
timeout 600 python bench_extras.py update_model 2>/dev/null | tail -1 | tee gpurun_out/${TAG}_update_model.json

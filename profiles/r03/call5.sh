#!/bin/bash
set -u
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
timeout 1500 python -m pytest tests/test_gpu_bench_launch_parity.py tests/test_gpu_meshes.py tests/test_gpu_baseline_sizes.py tests/test_gpu_vs_reference_device_kernels.py tests/test_gpu_reference_suite_replay.py tests/test_gpu_points_composite_interp.py -x -q -s 2>&1 | grep -E "^\[|passed|failed|Error|error" | cut -c1-700 | tail -40
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -3

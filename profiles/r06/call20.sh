#!/bin/bash
# round 6, call 20: the per-face record test + the suites that touch the backward
set -u
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_meshes.py tests/test_gpu_cover.py tests/test_gpu_bench_launch_parity.py tests/test_gpu_render_chain.py tests/test_gpu_baseline_sizes.py tests/test_gpu_pybind_boundary.py tests/test_gpu_world_transform.py tests/test_gpu_shading.py -x -q -m gpu 2>&1 | tail -n 4

#!/bin/bash
# round 6, call 44: backward suites on the build with seven waves per SIMD for the K = 4 / 8 per-face-record kernels
set -u
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
mkdir -p gpurun_out/r06c44
timeout 900 python -m pytest tests/test_gpu_meshes.py tests/test_gpu_cover.py tests/test_gpu_bench_launch_parity.py tests/test_gpu_baseline_sizes.py tests/test_gpu_render_chain.py tests/test_gpu_world_transform.py tests/test_gpu_pybind_boundary.py tests/test_gpu_reference_suite_replay.py -x -q -m gpu > gpurun_out/r06c44/t.txt 2>&1; tail -n 1 gpurun_out/r06c44/t.txt; grep -n "^E " gpurun_out/r06c44/t.txt | head

#!/bin/bash
set -u
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
timeout 300 python profiles/exp_measure.py 2>&1 | grep -v "^{" | tail -3
ABL_BATCH=64 timeout 300 python profiles/k_sweep.py 4 8 16 2>&1 | grep K=
timeout 900 python -m pytest tests/test_gpu_bench_launch_parity.py tests/test_gpu_meshes.py tests/test_gpu_vs_reference_device_kernels.py tests/test_gpu_reference_suite_replay.py tests/test_gpu_world_transform.py -x -q 2>&1 | tail -4

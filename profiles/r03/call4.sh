#!/bin/bash
set -u
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
O=gpurun_out/r03
mkdir -p $O
timeout 300 python profiles/exp_measure.py "$@" 2>&1 | grep -v "^{" | tail -8
timeout 900 python -m pytest tests/test_gpu_bench_launch_parity.py tests/test_gpu_meshes.py tests/test_gpu_reference_suite_replay.py tests/test_gpu_baseline_sizes.py -x -q 2>&1 | tail -5

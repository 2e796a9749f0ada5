#!/bin/bash
# round 5, call 19: the tile-sorted kernel as the binned K <= 16 path (register queues: naive launch only), point suites + sweep
set -u
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp OMP_NUM_THREADS=16
O=gpurun_out/r05c19
mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_points_composite_interp.py tests/test_gpu_vs_reference_device_kernels.py tests/test_gpu_baseline_sizes.py \
  tests/test_gpu_short_workspace.py tests/test_gpu_reference_suite_replay.py tests/test_gpu_points_renderer_dropin.py -x -q -p no:cacheprovider > $O/tests.txt 2>&1
echo "tests rc=$?"; tail -3 $O/tests.txt
timeout 200 python profiles/points_k_sweep.py 1 2 4 8 10 12 16 17 32 > $O/k_sweep.txt 2>&1; grep K= $O/k_sweep.txt

#!/bin/bash
# round 5, call 4: point_sorted_kernel with eight sub-batches in flight in pass A, prefetch in pass C, vector stores
set -u
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
O=gpurun_out/r05c4
mkdir -p $O
stamp() { echo "== $1 $(date +%T)" | tee -a $O/steps.txt; }
stamp tests
timeout 600 python -m pytest tests/test_gpu_points_composite_interp.py tests/test_gpu_vs_reference_device_kernels.py tests/test_gpu_baseline_sizes.py \
  tests/test_gpu_short_workspace.py tests/test_gpu_reference_suite_replay.py -x -q -p no:cacheprovider > $O/tests.txt 2>&1
echo "tests rc=$?"; tail -3 $O/tests.txt
stamp sweep_new
timeout 200 python profiles/points_k_sweep.py 1 4 8 10 16 32 50 64 100 150 > $O/k_sweep_new.txt 2>&1; cat $O/k_sweep_new.txt | grep K=
stamp chain
timeout 200 python -m pytest tests/test_gpu_points_renderer_dropin.py -q -s -p no:cacheprovider > $O/test_chain.txt 2>&1; tail -3 $O/test_chain.txt
grep -A12 worst_good_pixel $O/test_chain.txt | head -60
stamp end

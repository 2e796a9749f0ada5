#!/bin/bash
# round 5, call 28: CUDA tie order -- how many pixels the TIES kernels mark (the replay skipped), then the timing
set -u
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp OMP_NUM_THREADS=16
O=gpurun_out/r05c28
mkdir -p $O
timeout 300 python profiles/tie_order_timing.py --count-marks 8 4 16 2 > $O/tie_marks.txt 2>&1; tail -4 $O/tie_marks.txt
timeout 300 python profiles/tie_order_timing.py 8 1 4 16 > $O/tie_order_timing.txt 2>&1; tail -4 $O/tie_order_timing.txt
timeout 600 python -m pytest tests/test_gpu_vs_reference_device_kernels.py tests/test_gpu_short_workspace.py -x -q -p no:cacheprovider > $O/tests.txt 2>&1; tail -3 $O/tests.txt

#!/bin/bash
# round 5, call 32: the whole GPU suite on the build with the adjacent-lane flush, the cheap CUDA tie order and the binning scans
set -u
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp OMP_NUM_THREADS=16
O=gpurun_out/r05c32
mkdir -p $O
( time timeout 900 python -m pytest tests/ -x -q -m gpu -p no:cacheprovider ) > $O/tests.txt 2>&1; tail -6 $O/tests.txt
timeout 300 python profiles/tie_order_timing.py 8 > $O/tie_order_timing.txt 2>&1; tail -1 $O/tie_order_timing.txt

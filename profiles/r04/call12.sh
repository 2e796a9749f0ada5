#!/bin/bash
# Round 4, GPU call 12: the suites that had not seen the new binning (call 9 / 11) or the trimmed full-size oracle test yet.
set -u
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
O=gpurun_out/r04c12
mkdir -p $O
stamp() { echo "== $1 $(date +%T)" | tee -a $O/steps.txt; }
stamp shapes
timeout 200 python -m pytest tests/test_gpu_meshes.py -q -x -k "launch_shapes or coarse or empty" 2>&1 | tail -3 | tee $O/t_shapes.txt
stamp parity
timeout 300 python -m pytest tests/test_gpu_bench_launch_parity.py -q -x 2>&1 | tail -3 | tee $O/t_parity.txt
stamp refdev
timeout 400 python -m pytest tests/test_gpu_vs_reference_device_kernels.py -q -x 2>&1 | tail -3 | tee $O/t_refdev.txt
stamp sizes
timeout 600 python -m pytest tests/test_gpu_baseline_sizes.py -q -x --durations=5 2>&1 | tail -9 | tee $O/t_sizes.txt
stamp end

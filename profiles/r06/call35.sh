#!/bin/bash
# round 6, call 35: fused point chain with short workspaces; mesh suites on the split staging branch
set -u
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
mkdir -p gpurun_out/r06c35
timeout 600 python -m pytest tests/test_gpu_render_points.py tests/test_gpu_points_renderer_dropin.py -x -q -m gpu > gpurun_out/r06c35/t.txt 2>&1; tail -n 2 gpurun_out/r06c35/t.txt; grep -n "^E " gpurun_out/r06c35/t.txt | head

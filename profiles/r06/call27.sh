#!/bin/bash
# round 6, call 27: the cases of the patched PointsRenderer
set -u
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
mkdir -p gpurun_out/r06c27
timeout 600 python -m pytest tests/test_gpu_points_renderer_dropin.py -x -q -m gpu -k cases -s > gpurun_out/r06c27/cases.txt 2>&1; grep -n "^E \|passed\|failed\|Error" gpurun_out/r06c27/cases.txt | head -n 20; grep -n "^one_\|^three\|^back\|^padded\|^fallback" gpurun_out/r06c27/cases.txt | cut -c 1-400

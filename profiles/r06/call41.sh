#!/bin/bash
# round 6, call 41: the fused point chain against the reference-generated fixture
set -u
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
mkdir -p gpurun_out/r06c41
timeout 600 python -m pytest tests/test_gpu_render_chain.py -x -q -m gpu > gpurun_out/r06c41/t.txt 2>&1; tail -n 2 gpurun_out/r06c41/t.txt; grep -n "^E " gpurun_out/r06c41/t.txt | head

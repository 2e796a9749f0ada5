#!/bin/bash
# round 6, call 34: does the relative placement of the four output tensors move mesh_fine?
set -u
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
mkdir -p gpurun_out/r06c34
timeout 600 python profiles/placement_probe.py --iters 40 --rounds 2 > gpurun_out/r06c34/placement.txt 2> gpurun_out/r06c34/err.txt; tail -n 3 gpurun_out/r06c34/err.txt; tail -n 12 gpurun_out/r06c34/placement.txt | cut -c 1-200

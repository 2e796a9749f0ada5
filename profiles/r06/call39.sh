#!/bin/bash
# round 6, call 39b: outputs allocated FIRST in the process (before the inputs) vs later allocations
set -u
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
mkdir -p gpurun_out/r06c39
timeout 600 python profiles/placement_probe.py --iters 30 --rounds 1 > gpurun_out/r06c39/p_first.txt 2> gpurun_out/r06c39/err_first.txt
grep "^allocation\|^only the outputs" gpurun_out/r06c39/p_first.txt | cut -c 1-190; tail -n 3 gpurun_out/r06c39/err_first.txt

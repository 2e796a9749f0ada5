#!/bin/bash
# round 6, call 38b: is a slow allocation slow for a streaming fill too?
set -u
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
mkdir -p gpurun_out/r06c38
timeout 600 python profiles/placement_probe.py --iters 30 --rounds 1 > gpurun_out/r06c38/p_fill.txt 2> gpurun_out/r06c38/err_fill.txt
grep "^allocation\|^only the outputs\|^streaming" gpurun_out/r06c38/p_fill.txt | cut -c 1-170

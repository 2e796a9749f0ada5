#!/bin/bash
# round 6, call 26: the whole GPU suite on the build with the fused PointsRenderer chain and the per-face records
set -u
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
O=gpurun_out/r06c26
mkdir -p $O
( time timeout 1100 python -m pytest tests/ -x -q -m gpu -p no:cacheprovider ) > $O/tests_full.txt 2>&1; echo "tests rc=$?" | tee -a $O/tests_full.txt
grep -n "^E \|passed\|failed\|^real" $O/tests_full.txt | tail -n 15

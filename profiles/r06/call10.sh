#!/bin/bash
# round 6, call 10: pybind boundary tests, cover tests, bench contract tests on the rebuilt tree
set -u
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
O=gpurun_out/r06c10
mkdir -p $O
( time timeout 900 python -m pytest tests/test_gpu_pybind_boundary.py tests/test_gpu_cover.py tests/test_gpu_bench_contract.py -x -q -s -p no:cacheprovider ) > $O/tests.txt 2>&1; echo "tests rc=$?"
grep -n "config 2 per call" $O/tests.txt; tail -4 $O/tests.txt

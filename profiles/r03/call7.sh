#!/bin/bash
set -u
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
mkdir -p gpurun_out/r03
echo "=== _C only"; timeout 1200 python tests/run_reference_suite.py --out gpurun_out/r03/ref_suite_c_only.json 2>&1 | tail -40
echo "=== patched"; timeout 1200 python tests/run_reference_suite.py --patch-python --out gpurun_out/r03/ref_suite_patched.json 2>&1 | tail -60

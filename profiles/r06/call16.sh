#!/bin/bash
# round 6, call 16: list by plan rank -- cover / mesh suites, then the cover list on / off through bench.py
set -u
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
O=gpurun_out/r06c16
mkdir -p $O
( time timeout 900 python -m pytest tests/test_gpu_cover.py tests/test_gpu_meshes.py tests/test_gpu_bench_launch_parity.py -x -q -p no:cacheprovider ) > $O/tests.txt 2>&1; echo "tests rc=$?"
tail -3 $O/tests.txt
sed -i 's#gpurun_out/r06c15#gpurun_out/r06c16#g' profiles/r06/call15.sh
bash profiles/r06/call15.sh

#!/bin/bash
# round 6, call 6: variant(s) against the product in one process, then tests/test_gpu_meshes.py on the first variant
set -u
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
O=gpurun_out/r06c6
mkdir -p $O
L=$PWD/pytorch3d_amd
SPEC=""
for v in "$@"; do SPEC="$SPEC $v=$L/libp3d_$v.so"; done
timeout 300 python profiles/exp_measure.py --iters 100 $SPEC > $O/exp_measure.jsonl 2> $O/exp_measure.txt; tail -n $(( $# + 2 )) $O/exp_measure.txt
P3D_LIB_PATH=$L/libp3d_$1.so timeout 400 python -m pytest tests/test_gpu_meshes.py tests/test_gpu_bench_launch_parity.py tests/test_gpu_vs_reference_device_kernels.py -x -q > $O/tests_$1.txt 2>&1; tail -3 $O/tests_$1.txt

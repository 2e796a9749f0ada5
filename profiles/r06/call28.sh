#!/bin/bash
# round 6, call 28: the 32 reads of the pixel-centre table hoisted in front of the staging branch (safe under spills by construction) vs the product
set -u
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
O=gpurun_out/r06c28
mkdir -p $O
L=$PWD/pytorch3d_amd
for i in 1 2; do
timeout 300 python profiles/exp_measure.py --iters 100 hoist=$L/libp3d_hoist.so > $O/exp_$i.jsonl 2> $O/exp_$i.txt; tail -n 3 $O/exp_$i.txt
done
P3D_LIB_PATH=$L/libp3d_hoist.so timeout 400 python -m pytest tests/test_gpu_bench_launch_parity.py tests/test_gpu_cover.py tests/test_gpu_meshes.py tests/test_gpu_vs_reference_device_kernels.py -x -q > $O/tests_hoist.txt 2>&1; tail -n 1 $O/tests_hoist.txt

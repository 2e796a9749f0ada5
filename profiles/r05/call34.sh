#!/bin/bash
# round 5, call 34: the backward's edge-ranking distances with mul + add fused (seg_dist2_rank) against the product
set -u
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp OMP_NUM_THREADS=16
O=gpurun_out/r05c34
mkdir -p $O
V=$PWD/pytorch3d_amd/libp3d_rank.so
for td in 1.0 1.5; do
  timeout 300 python profiles/exp_measure.py --iters 40 --torus-div $td rank=$V > $O/exp_measure_$td.json 2> $O/exp_measure_$td.txt; tail -3 $O/exp_measure_$td.txt
done
P3D_LIB_PATH=$V timeout 600 python -m pytest tests/test_gpu_meshes.py tests/test_gpu_bench_launch_parity.py tests/test_gpu_baseline_sizes.py tests/test_gpu_vs_reference_device_kernels.py -x -q -p no:cacheprovider > $O/tests.txt 2>&1; tail -3 $O/tests.txt

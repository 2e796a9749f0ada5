#!/bin/bash
# round 6, call 33: like for like (all as variants, two orders): the table read inside the branch (nohoist), in front of it (same = product), between the two halves of the branch under a scalar guard (split)
set -u
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
O=gpurun_out/r06c33
mkdir -p $O
L=$PWD/pytorch3d_amd
for i in 1 2; do
timeout 400 python profiles/exp_measure.py --iters 100 split=$L/libp3d_split.so nohoist=$L/libp3d_nohoist.so same=$L/libp3d_same.so > $O/a_$i.jsonl 2> $O/a_$i.txt; tail -n 5 $O/a_$i.txt
timeout 400 python profiles/exp_measure.py --iters 100 same=$L/libp3d_same.so nohoist=$L/libp3d_nohoist.so split=$L/libp3d_split.so > $O/b_$i.jsonl 2> $O/b_$i.txt; tail -n 5 $O/b_$i.txt
done
P3D_LIB_PATH=$L/libp3d_split.so timeout 400 python -m pytest tests/test_gpu_bench_launch_parity.py tests/test_gpu_cover.py tests/test_gpu_meshes.py tests/test_gpu_vs_reference_device_kernels.py -x -q > $O/tests_split.txt 2>&1; tail -n 1 $O/tests_split.txt

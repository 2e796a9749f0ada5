#!/bin/bash
# round 5, call 20: where the tile-sorted kernel stops paying against the single-wave sorted kernel (variant: tile-sorted up to K = 64)
set -u
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp OMP_NUM_THREADS=16
O=gpurun_out/r05c20
mkdir -p $O
echo "[tile-sorted up to 64]"; P3D_LIB_PATH=$PWD/pytorch3d_amd/libp3d_ts64.so timeout 200 python profiles/points_k_sweep.py 16 17 20 24 28 32 40 48 64 > $O/k_sweep_ts64.txt 2>&1; grep K= $O/k_sweep_ts64.txt
echo "[product: sorted kernel above 16]"; timeout 200 python profiles/points_k_sweep.py 16 17 20 24 28 32 40 48 64 > $O/k_sweep_product.txt 2>&1; grep K= $O/k_sweep_product.txt
P3D_LIB_PATH=$PWD/pytorch3d_amd/libp3d_ts64.so timeout 300 python -m pytest tests/test_gpu_points_composite_interp.py -x -q -p no:cacheprovider 2>&1 | tail -2

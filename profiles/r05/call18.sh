#!/bin/bash
# round 5, calls 16 and 18: point_tile_sorted_kernel (cooperative workgroup + exact order + LDS append queues, K <= 16) against the register queues
# with the tile pre-sort (libp3d_r5queues.so = the same sources with -DP3D_TILE_SORTED_MAX_K=0)
set -u
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp OMP_NUM_THREADS=16
O=gpurun_out/r05c18
mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_points_composite_interp.py tests/test_gpu_vs_reference_device_kernels.py tests/test_gpu_baseline_sizes.py \
  tests/test_gpu_short_workspace.py tests/test_gpu_reference_suite_replay.py tests/test_gpu_points_renderer_dropin.py -x -q -p no:cacheprovider > $O/tests.txt 2>&1
echo "tests rc=$?"; tail -3 $O/tests.txt
echo "[tile-sorted]"; timeout 200 python profiles/points_k_sweep.py 1 2 4 8 10 12 16 > $O/k_sweep_tile_sorted.txt 2>&1; grep K= $O/k_sweep_tile_sorted.txt
echo "[queues + pre-sort]"; P3D_LIB_PATH=$PWD/pytorch3d_amd/libp3d_r5queues.so timeout 200 python profiles/points_k_sweep.py 1 2 4 8 10 12 16 > $O/k_sweep_queues.txt 2>&1; grep K= $O/k_sweep_queues.txt

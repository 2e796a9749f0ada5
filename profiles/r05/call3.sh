#!/bin/bash
# round 5, call 3: point_sorted_kernel (one wave per 8x8 sub-tile, exact front-to-back order, LDS append queues) against the
# round-4 queue kernels (libp3d_r5base.so = HEAD before the change): parity suites on the new library, K sweep on both, and the
# PointsRenderer drop-in in both shim modes (+ the OMP_NUM_THREADS question of call 2: 25 ms per step inside bench.py).
set -u
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
O=gpurun_out/r05c3
mkdir -p $O
stamp() { echo "== $1 $(date +%T)" | tee -a $O/steps.txt; }
stamp tests
timeout 600 python -m pytest tests/test_gpu_points_composite_interp.py tests/test_gpu_vs_reference_device_kernels.py tests/test_gpu_baseline_sizes.py \
  tests/test_gpu_short_workspace.py tests/test_gpu_reference_suite_replay.py tests/test_gpu_points_renderer_dropin.py -x -q -p no:cacheprovider > $O/tests.txt 2>&1
echo "tests rc=$?"; tail -4 $O/tests.txt
stamp sweep_new
timeout 200 python profiles/points_k_sweep.py 1 4 8 10 16 32 50 64 100 150 > $O/k_sweep_new.txt 2>&1; cat $O/k_sweep_new.txt | grep K=
stamp sweep_base
P3D_LIB_PATH=$PWD/pytorch3d_amd/libp3d_r5base.so timeout 300 python profiles/points_k_sweep.py 1 4 8 10 16 32 50 64 100 150 > $O/k_sweep_base.txt 2>&1; cat $O/k_sweep_base.txt | grep K=
stamp dropin
for m in c_only patched; do
  OMP_NUM_THREADS=16 timeout 120 python profiles/dropin_points_timing.py --mode $m > $O/dropin_points_${m}_omp16.json 2>$O/dropin_points_${m}.err
  python -c "import json;j=json.load(open('$O/dropin_points_${m}_omp16.json'));print('$m omp16', j['ms_per_step'], j['our_kernels_sum_ms'], j.get('patched_calls'))"
done
env -u OMP_NUM_THREADS timeout 120 python profiles/dropin_points_timing.py --mode c_only > $O/dropin_points_c_only_noomp.json 2>/dev/null
python -c "import json;j=json.load(open('$O/dropin_points_c_only_noomp.json'));print('c_only no OMP_NUM_THREADS', j['ms_per_step'])"
stamp end

#!/bin/bash
# round 5, call 5: tile pre-sort in the queue kernels (K <= 32 in this build) vs the sorted kernel for every K (libp3d_r5sorted.so)
set -u
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
O=gpurun_out/r05c5
mkdir -p $O
stamp() { echo "== $1 $(date +%T)" | tee -a $O/steps.txt; }
stamp tests
timeout 600 python -m pytest tests/test_gpu_points_composite_interp.py tests/test_gpu_vs_reference_device_kernels.py tests/test_gpu_baseline_sizes.py \
  tests/test_gpu_short_workspace.py tests/test_gpu_reference_suite_replay.py -x -q -p no:cacheprovider > $O/tests.txt 2>&1
echo "tests rc=$?"; tail -3 $O/tests.txt
stamp sweep_presort_queues
timeout 200 python profiles/points_k_sweep.py 1 4 8 10 12 16 24 32 40 > $O/k_sweep_presort.txt 2>&1; cat $O/k_sweep_presort.txt | grep K=
stamp sweep_sorted
P3D_LIB_PATH=$PWD/pytorch3d_amd/libp3d_r5sorted.so timeout 200 python profiles/points_k_sweep.py 1 4 8 10 12 16 24 32 40 > $O/k_sweep_sorted.txt 2>&1; cat $O/k_sweep_sorted.txt | grep K=
stamp chain
timeout 200 python -m pytest tests/test_gpu_points_renderer_dropin.py -q -s -p no:cacheprovider > $O/test_chain.txt 2>&1; tail -3 $O/test_chain.txt
python - <<'PY'
import re,json
t=open('gpurun_out/r05c5/test_chain.txt').read()
i=t.find('"worst_good_pixel"')
print(t[i:i+2500].replace('\n',' ').replace('   ',' ')[:1800])
PY
stamp end

#!/bin/bash
# round 5, call 36: list entries of (id, depth bits) written by bin_fill with one 8-byte store (call 35: a second array of depths
# doubled bin_fill's store requests, 0.024 -> 0.046 ms, for 0.110 -> 0.095 in points_fine): config 4, K sweep, the point suites
set -u
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp OMP_NUM_THREADS=16
O=gpurun_out/r05c36
mkdir -p $O
for rep in 1 2; do
  timeout 200 python profiles/dropin_points_timing.py --mode patched --steps 50 > $O/points_$rep.json 2>&1
  python - <<PY
import json
j=json.loads([l for l in open("$O/points_$rep.json") if l.startswith("{")][-1])
k=j["our_kernels_ms_per_step"]
print(round(j["ms_per_step"],4), k, "sum", j["our_kernels_sum_ms"])
PY
done
timeout 200 python profiles/points_k_sweep.py 1 8 10 16 24 28 32 64 > $O/k_sweep.txt 2>&1; grep K= $O/k_sweep.txt
timeout 600 python -m pytest tests/test_gpu_points_composite_interp.py tests/test_gpu_points_renderer_dropin.py tests/test_gpu_short_workspace.py tests/test_gpu_vs_reference_device_kernels.py tests/test_gpu_baseline_sizes.py tests/test_gpu_reference_suite_replay.py -x -q -p no:cacheprovider > $O/tests.txt 2>&1; tail -3 $O/tests.txt

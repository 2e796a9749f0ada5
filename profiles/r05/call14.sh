#!/bin/bash
# round 5, call 14: a point's channels with one 12-byte request in the compositors (renderers' (P, 3) features)
set -u
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp OMP_NUM_THREADS=16
O=gpurun_out/r05c14
mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_points_composite_interp.py tests/test_gpu_vs_reference_device_kernels.py tests/test_gpu_reference_suite_replay.py tests/test_gpu_points_renderer_dropin.py tests/test_gpu_render_chain.py -x -q -p no:cacheprovider > $O/tests.txt 2>&1
echo "tests rc=$?"; tail -3 $O/tests.txt
for m in c_only patched; do
  timeout 120 python profiles/dropin_points_timing.py --mode $m --steps 50 > $O/dropin_points_$m.json 2>/dev/null
  python -c "import json;j=json.load(open('$O/dropin_points_$m.json'));print('points $m', round(j['ms_per_step'],4), j['our_kernels_sum_ms'], j['our_kernels_ms_per_step'])"
done

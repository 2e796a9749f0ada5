#!/bin/bash
# round 5, call 13: the compositors on the renderers' own feature layout ((P, C) rows through strides), through the drop-in chain
set -u
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp OMP_NUM_THREADS=16
O=gpurun_out/r05c13
mkdir -p $O
for m in c_only patched; do
  timeout 120 python profiles/dropin_points_timing.py --mode $m --steps 50 > $O/dropin_points_$m.json 2>/dev/null
  python -c "import json;j=json.load(open('$O/dropin_points_$m.json'));print('points $m', round(j['ms_per_step'],4), j['our_kernels_sum_ms'], j['our_kernels_ms_per_step'])"
done
timeout 200 python -m pytest tests/test_gpu_points_renderer_dropin.py tests/test_gpu_reference_own_tests.py -x -q -p no:cacheprovider 2>&1 | tail -3

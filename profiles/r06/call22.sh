#!/bin/bash
# round 6, call 22: the fused PointsRenderer chain (p3d_rasterize_points_composite + _backward): its tests, the points suites, the drop-in
set -u
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
O=gpurun_out/r06c22
mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_render_points.py -x -q -m gpu > $O/render.txt 2>&1; tail -n 3 $O/render.txt; grep -n "^E " $O/render.txt | head -n 20
timeout 900 python -m pytest tests/test_gpu_points_composite_interp.py tests/test_gpu_points_renderer_dropin.py -x -q -m gpu > $O/points.txt 2>&1; tail -n 3 $O/points.txt; grep -n "^E " $O/points.txt | head -n 20
for m in "--mode patched" "--mode patched --no-fuse" "--mode c_only"; do
timeout 300 python profiles/dropin_points_timing.py $m --steps 50 2>$O/err.txt | tail -n 1 > $O/t.json
python - <<PY
import json
j=json.load(open('$O/t.json'))
print(j['mode'], 'ms/step', round(j['ms_per_step'],4), 'kernels', j['our_kernels_sum_ms'], j['our_kernels_ms_per_step'])
PY
cp $O/t.json "$O/timing_$(echo $m | tr -d ' -').json"
done

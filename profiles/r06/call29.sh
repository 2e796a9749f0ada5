#!/bin/bash
# round 6, call 29: the row scan with the offsets scan as its tail (one image, many chunks): launch-shape tests, point suites, config-4 timing
set -u
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
O=gpurun_out/r06c29
mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_meshes.py -x -q -m gpu -k "launch_shapes" > $O/shapes.txt 2>&1; tail -n 2 $O/shapes.txt; grep -n "^E " $O/shapes.txt | head -n 10
timeout 900 python -m pytest tests/test_gpu_points_composite_interp.py tests/test_gpu_render_points.py tests/test_gpu_points_renderer_dropin.py tests/test_gpu_short_workspace.py tests/test_gpu_baseline_sizes.py -x -q -m gpu > $O/points.txt 2>&1; tail -n 2 $O/points.txt; grep -n "^E " $O/points.txt | head -n 10
for i in 1 2 3; do
timeout 300 python profiles/dropin_points_timing.py --mode patched --steps 100 2>$O/err.txt | tail -n 1 > $O/t_$i.json
python - <<PY
import json
j=json.load(open('$O/t_$i.json'))
print(j['mode'], 'ms/step', round(j['ms_per_step'],4), 'kernels', j['our_kernels_sum_ms'], j['our_kernels_ms_per_step'])
PY
done

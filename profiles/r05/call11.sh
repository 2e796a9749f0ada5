#!/bin/bash
# round 5, calls 11-12: compositors read sample-major inputs 64 consecutive entries per instruction (LDS transpose); features through their strides ((P, C) rows: one request per point)
set -u
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp OMP_NUM_THREADS=16
O=gpurun_out/r05c12
mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_points_composite_interp.py tests/test_gpu_vs_reference_device_kernels.py tests/test_gpu_reference_suite_replay.py tests/test_gpu_points_renderer_dropin.py tests/test_gpu_render_chain.py -x -q -p no:cacheprovider > $O/tests.txt 2>&1
echo "tests rc=$?"; tail -3 $O/tests.txt
timeout 300 python bench.py --steps 20 --no-cpu-baseline --no-dropin > $O/bench.json 2>$O/bench.err
python - <<'PY'
import json
b=json.load(open('gpurun_out/r05c12/bench.json'))
print(b['value'], b['ms_per_step'])
for k,v in b['other_configs'].items(): print(k, v.get('wall_ms'), v.get('kernel_sum_ms'), v.get('kernels_ms'))
PY

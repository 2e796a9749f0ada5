#!/bin/bash
# round 6, call 30: the scan tail's race test + the patched PointsRenderer cases on the exact-type list path
set -u
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
O=gpurun_out/r06c30
mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_points_composite_interp.py -x -q -m gpu -k "race_free" > $O/race.txt 2>&1; tail -n 2 $O/race.txt; grep -n "^E " $O/race.txt | head
timeout 600 python -m pytest tests/test_gpu_points_renderer_dropin.py -x -q -m gpu > $O/dropin.txt 2>&1; tail -n 2 $O/dropin.txt; grep -n "^E " $O/dropin.txt | head
timeout 900 python bench.py --steps 50 --no-cpu-baseline --no-reference-device --no-dropin > $O/bench.json 2> $O/bench.err
python - <<'PY'
import json
b=json.loads([l for l in open('gpurun_out/r06c30/bench.json') if l.startswith('{')][0])
print(round(b['value'],1), b['kernels_ms'])
for k in ('config4_points_1m_512_k10_fwd_bwd','config4_points_1m_512_k10_fwd_bwd_fused','config2_cow_256_k8_fwd'):
    v=b['other_configs'].get(k,{})
    print(k, v.get('wall_ms'), v.get('kernel_sum_ms'), v.get('kernels_ms'))
PY

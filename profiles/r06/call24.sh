#!/bin/bash
# round 6, call 24: python bench.py with all legs on the build that carries the fused PointsRenderer chain + per-face records
set -u
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
O=gpurun_out/r06c24
mkdir -p $O
timeout 1500 python bench.py > $O/bench.json 2> $O/bench.err; tail -n 3 $O/bench.err
python - <<'PY'
import json
b=json.loads([l for l in open('gpurun_out/r06c24/bench.json') if l.startswith('{')][0])
print(round(b['value'],1), b['unit'], b['ms_per_step'], 'fine', b['kernels_ms'].get('mesh_fine'), 'bwd', b['kernels_ms'].get('mesh_backward'))
oc=b['other_configs']
for k in ('config4_points_1m_512_k10_fwd_bwd','config4_points_1m_512_k10_fwd_bwd_fused'):
    v=oc.get(k,{})
    print(k, v.get('wall_ms'), v.get('kernel_sum_ms'), v.get('kernels_ms'), v.get('vs_operator_chain'), v.get('error'))
d=oc.get('config4_points_renderer_dropin',{})
for m,v in d.items():
    print(m, v.get('ms_per_step'), v.get('our_kernels_sum_ms'), v.get('reason'))
PY

#!/bin/bash
# round 6, call 19: per-face reciprocals for the backward (p3d_gather_face_verts_pre -> mesh_backward PRE): bench with / without, then parity
set -u
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
O=gpurun_out/r06c19
mkdir -p $O
for i in 1 2; do
for L in 1 0; do
P3D_FACE_PRE=$L timeout 600 python bench.py --steps 100 --warmup 5 --no-cpu-baseline --no-dropin --no-reference-device --no-other-configs > $O/bench_${L}_$i.json 2> $O/bench.err
python - <<PY
import json
b=json.loads([l for l in open('$O/bench_${L}_$i.json') if l.startswith('{')][0])
print('bench pre=$L', round(b['value'],1),'Mpix/s',round(b['ms_per_step'],4),'ms', 'fine', b['kernels_ms']['mesh_fine'], 'bwd', b['kernels_ms']['mesh_backward'], 'gather', b['kernels_ms'].get('gather_face_verts'))
PY
done
done
timeout 900 python -m pytest tests/test_gpu_meshes.py tests/test_gpu_cover.py tests/test_gpu_bench_launch_parity.py -x -q -m gpu 2>&1 | tail -n 8

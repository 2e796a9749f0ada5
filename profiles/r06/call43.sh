#!/bin/bash
# round 6, call 43: the light batch (tori / 1.5) and config 2 with the backward at seven vs six waves per SIMD
set -u
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
O=gpurun_out/r06c43
mkdir -p $O
L=$PWD/pytorch3d_amd
for i in 1 2; do
for v in amd occ7; do
P3D_LIB_PATH=$L/libp3d_$v.so timeout 600 python bench.py --steps 50 --warmup 5 --no-cpu-baseline --no-dropin --no-reference-device > $O/bench_${v}_$i.json 2> $O/bench.err
python - <<PY
import json
b=json.loads([l for l in open('$O/bench_${v}_$i.json') if l.startswith('{')][0])
lw=b.get('workload_torus_div_1.5',{})
print('$v', round(b['value'],1), 'bwd', b['kernels_ms']['mesh_backward'], 'light batch:', lw.get('value'), lw.get('ms_per_step'), lw.get('kernels_ms'))
PY
done
done

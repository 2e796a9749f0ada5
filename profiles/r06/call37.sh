#!/bin/bash
# round 6, call 37: points_fine's epilogue: fragments and image in one loop (product) vs fragments first, image in a second walk
set -u
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
O=gpurun_out/r06c37
mkdir -p $O
L=$PWD/pytorch3d_amd
for i in 1 2 3; do
for v in amd twophase; do
P3D_LIB_PATH=$L/libp3d_$v.so timeout 300 python profiles/dropin_points_timing.py --mode patched --steps 100 2>$O/err.txt | tail -n 1 > $O/t.json
python - <<PY
import json
j=json.load(open('$O/t.json'))
print('$v', 'ms/step', round(j['ms_per_step'],4), 'points_fine', j['our_kernels_ms_per_step']['points_fine'], 'sum', j['our_kernels_sum_ms'])
PY
done
done

#!/bin/bash
# round 6, call 45: the reference-signature backward with face records: mesh suites, the reference's own tests over _C alone, drop-in timing
set -u
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
O=gpurun_out/r06c45
mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_meshes.py tests/test_gpu_cover.py tests/test_gpu_bench_launch_parity.py tests/test_gpu_reference_own_tests.py tests/test_gpu_pybind_boundary.py tests/test_gpu_clip.py -x -q -m gpu > $O/t.txt 2>&1; tail -n 1 $O/t.txt; grep -n "^E " $O/t.txt | head
for P in 1 0; do
P3D_FACE_PRE=$P timeout 300 python profiles/dropin_timing.py --mode c_only 2>$O/err.txt | tail -n 1 > $O/c_$P.json
python - <<PY
import json
j=json.load(open('$O/c_$P.json'))
print('c_only FACE_PRE=$P', round(j['ms_per_step'],3), j['our_kernels_ms_per_step'])
PY
done

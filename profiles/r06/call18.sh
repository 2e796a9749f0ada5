#!/bin/bash
# round 6, call 18: config 4 through the patched PointsRenderer, eager and replayed from a HIP graph
set -u
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp OMP_NUM_THREADS=16
O=gpurun_out/r06c18
mkdir -p $O
timeout 300 python -X faulthandler profiles/dropin_points_timing.py --mode patched --steps 10 > $O/c4_patched.json 2> $O/err0.txt; echo rc=$?; tail -12 $O/err0.txt
timeout 300 python -X faulthandler profiles/dropin_points_timing.py --mode patched --graph --steps 30 > $O/c4_patched_graph.json 2> $O/err.txt; echo rc=$?
tail -25 $O/err.txt
python - <<'PY'
import json
for f in ('c4_patched.json','c4_patched_graph.json'):
    l=[l for l in open('gpurun_out/r06c18/'+f) if l.startswith('{')]
    if l:
        b=json.loads(l[-1]); print(f, {k:b.get(k) for k in ('ms_per_step','our_kernels_sum_ms','graph','value','reason')})
PY

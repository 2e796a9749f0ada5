#!/bin/bash
# round 6, call 15: the cover list on / off through bench.py, alternating, 100 steps each
set -u
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
O=gpurun_out/r06c15
mkdir -p $O
for i in 1 2; do
for L in 1 0; do
P3D_COVER_LIST=$L timeout 600 python bench.py --steps 100 --warmup 5 --no-cpu-baseline --no-dropin --no-reference-device --no-other-configs > $O/bench_${L}_$i.json 2> $O/bench.err
python - <<PY
import json
b=json.loads([l for l in open('gpurun_out/r06c15/bench_${L}_$i.json') if l.startswith('{')][0])
print('list=$L', round(b['value'],1),'Mpix/s',round(b['ms_per_step'],4),'ms', b['kernels_ms'])
PY
done
done

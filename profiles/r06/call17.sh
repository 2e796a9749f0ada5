#!/bin/bash
# round 6, call 17: the same build through the two harnesses on ONE box: exp_measure (operators back to back over the C ABI) and bench.py
set -u
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
O=gpurun_out/r06c17
mkdir -p $O
for i in 1 2; do
timeout 300 python profiles/exp_measure.py --iters 100 > $O/exp_$i.jsonl 2> $O/exp_$i.txt; tail -n 2 $O/exp_$i.txt
for L in 1 0; do
P3D_COVER_LIST=$L timeout 600 python bench.py --steps 100 --warmup 5 --no-cpu-baseline --no-dropin --no-reference-device --no-other-configs > $O/bench_${L}_$i.json 2> $O/bench.err
python - <<PY
import json
b=json.loads([l for l in open('$O/bench_${L}_$i.json') if l.startswith('{')][0])
print('bench list=$L', round(b['value'],1),'Mpix/s',round(b['ms_per_step'],4),'ms', 'fine', b['kernels_ms']['mesh_fine'], 'bwd', b['kernels_ms']['mesh_backward'])
PY
done
done

#!/bin/bash
# round 5, call 15: the driver's exact command (`--gpus 1 --steps 20 --warmup 5`) with and without the untimed pre-warm, three times each
set -u
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
O=gpurun_out/r05c15
mkdir -p $O
for i in 1 2 3; do
  for pw in 0 0.4; do
    timeout 300 python3 bench.py --gpus 1 --steps 20 --warmup 5 --prewarm-s $pw --no-cpu-baseline --no-other-configs --no-dropin > $O/bench_pw${pw}_$i.json 2>/dev/null
    python -c "import json;b=json.load(open('$O/bench_pw${pw}_$i.json'));print('prewarm $pw run $i:', round(b['value'],1), 'Mpix/s', round(b['ms_per_step'],4), 'median', round(b['ms_per_step_median'],4), b['kernels_ms']['mesh_fine'], b['kernels_ms']['mesh_backward'])"
  done
done

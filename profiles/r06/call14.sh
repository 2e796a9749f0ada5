#!/bin/bash
# round 6, call 14: cover list -- cover / mesh / launch-parity suites on the product, then the driver-form bench
set -u
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
O=gpurun_out/r06c14
mkdir -p $O
( time timeout 900 python -m pytest tests/test_gpu_cover.py tests/test_gpu_meshes.py tests/test_gpu_bench_launch_parity.py tests/test_gpu_baseline_sizes.py -x -q -p no:cacheprovider ) > $O/tests.txt 2>&1; echo "tests rc=$?"
tail -3 $O/tests.txt
for i in 1 2; do
timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-dropin --no-reference-device --no-other-configs > $O/bench_$i.json 2> $O/bench.err; echo "bench rc=$?"
python - <<PY
import json
b=json.loads([l for l in open('gpurun_out/r06c14/bench_$i.json') if l.startswith('{')][0])
print(round(b['value'],1),'Mpix/s',round(b['ms_per_step'],4),'ms; without prewarm', round(b['without_prewarm']['value'],1), b['kernels_ms'])
PY
done

#!/bin/bash
# round 6, call 9: bench contract + cover tests on the rebuilt library; the driver-form bench (20 steps, 5 warm-up); rocprofv3
# kernel trace + PMC passes of the bench step (profiles/run_rocprof.sh)
set -u
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
O=gpurun_out/r06c9
mkdir -p $O
( time timeout 900 python -m pytest tests/test_gpu_bench_contract.py tests/test_gpu_cover.py tests/test_gpu_meshes.py -x -q -p no:cacheprovider ) > $O/tests.txt 2>&1; echo "tests rc=$?"
tail -4 $O/tests.txt
timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-dropin --no-reference-device > $O/bench_driver_form.json 2> $O/bench.err; echo "bench rc=$?"
python - <<'PY'
import json
b=json.loads([l for l in open('gpurun_out/r06c9/bench_driver_form.json') if l.startswith('{')][0])
print(round(b['value'],1),'Mpix/s',round(b['ms_per_step'],4),'ms; without prewarm', b.get('without_prewarm'), b['kernels_ms'])
PY
timeout 700 bash profiles/run_rocprof.sh $O/prof > $O/rocprof.log 2>&1; tail -3 $O/rocprof.log

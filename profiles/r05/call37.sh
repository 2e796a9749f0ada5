#!/bin/bash
# round 5, call 37: the whole GPU suite, smoke() and the driver's form of the bench on the last build (wave_table.h: the two
# flushes as #ifdef / #else branches; no behaviour change)
set -u
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
O=gpurun_out/r05c37
mkdir -p $O
( time timeout 900 python -m pytest tests/ -x -q -m gpu -p no:cacheprovider ) > $O/tests.txt 2>&1; tail -5 $O/tests.txt
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.txt 2>&1; tail -1 $O/smoke.txt
timeout 600 python3 bench.py --gpus 1 --steps 20 --warmup 5 > $O/driver_form.json 2> $O/driver_form.err
python3 - <<PY
import json
j=json.loads(open("$O/driver_form.json").read().strip().splitlines()[-1])
print(round(j["value"],1), round(j["ms_per_step"],4), j["roofline"]["frac"], j["kernels_ms"], j["config"]["dropin_ms_per_step"])
c=j["other_configs"]["config4_points_renderer_dropin"]
print({m:(round(c[m]["ms_per_step"],3), c[m]["our_kernels_sum_ms"]) for m in c})
PY

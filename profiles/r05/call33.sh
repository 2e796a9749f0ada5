#!/bin/bash
# round 5, call 33: the driver's form of the bench on the closing build, three times
set -u
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
O=gpurun_out/r05c33
mkdir -p $O
for i in 1 2 3; do
  timeout 600 python3 bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-other-configs --no-dropin > $O/driver_form_$i.json 2> $O/driver_form_$i.err
  python3 - <<PY
import json
j=json.loads(open("$O/driver_form_$i.json").read().strip().splitlines()[-1])
print($i, round(j["value"],1), round(j["ms_per_step"],4), j["roofline"]["frac"], j["roofline"]["per_kernel"]["mesh_backward"]["avg_ms"])
PY
done
timeout 600 python3 bench.py --gpus 1 --steps 20 --warmup 5 > $O/driver_form_full.json 2> $O/driver_form_full.err
python3 - <<PY
import json
j=json.loads(open("$O/driver_form_full.json").read().strip().splitlines()[-1])
print("full", round(j["value"],1), round(j["ms_per_step"],4), j["roofline"]["frac"], j["config"]["dropin_ms_per_step"])
PY

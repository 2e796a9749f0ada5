#!/bin/bash
# round 6, call 7: the whole GPU suite + smoke + a short bench on the committed build
set -u
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
O=gpurun_out/r06c7
mkdir -p $O
( time timeout 1100 python -m pytest tests/ -x -q -m gpu -p no:cacheprovider ) > $O/tests_full.txt 2>&1; echo "tests rc=$?" | tee -a $O/tests_full.txt
tail -4 $O/tests_full.txt
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.txt 2>&1; echo "smoke rc=$?"; tail -1 $O/smoke.txt
timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-dropin > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"
python -c "
import json;b=json.load(open('$O/bench.json'));print(round(b['value'],1), 'Mpix/s', round(b['ms_per_step'],4), 'ms', b['kernels_ms'])"

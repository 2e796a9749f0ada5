#!/bin/bash
# round 6, evidence on the round's build: the whole `pytest -m gpu` suite (the driver's command) + smoke(), `python bench.py` (the
# driver's command, every leg), the driver-form bench (20 steps, 5 warm-up), rocprofv3 kernel trace + PMC passes of the bench step.
set -u
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
O=gpurun_out/r06final
mkdir -p $O
stamp() { echo "== $1 $(date +%T)" | tee -a $O/steps.txt; }
stamp tests
( time timeout 1100 python -m pytest tests/ -x -q -m gpu -p no:cacheprovider ) > $O/tests_full.txt 2>&1; echo "tests rc=$?" | tee -a $O/tests_full.txt
tail -4 $O/tests_full.txt
stamp smoke
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.txt 2>&1; echo "smoke rc=$?"; tail -1 $O/smoke.txt
stamp bench
timeout 900 python bench.py > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"
cut -c1-300 $O/bench.json
stamp bench_driver_form
timeout 600 python bench.py --steps 20 --warmup 5 > $O/bench_driver_form.json 2> $O/bench_driver_form.err; echo "bench rc=$?"
cut -c1-200 $O/bench_driver_form.json
stamp rocprof_step
timeout 700 bash profiles/run_rocprof.sh $O/prof > $O/rocprof.log 2>&1
du -sh $O
stamp end

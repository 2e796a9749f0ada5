#!/bin/bash
# round 5, evidence on the round's build: the whole `pytest -m gpu` suite (the driver's command) + smoke(), `python bench.py` (the
# driver's command), rocprofv3 kernel trace + PMC passes of (a) the bench step, (b) BASELINE configs[3] through the patched
# PointsRenderer drop-in, (c) the soft-Phong pipeline (SURVEY 8(f) kernels).
set -u
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
O=gpurun_out/r05final
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
stamp rocprof_step
timeout 500 bash profiles/run_rocprof.sh $O/prof > $O/rocprof.log 2>&1
stamp rocprof_c4
timeout 400 bash profiles/run_rocprof.sh $O/prof_c4 "python profiles/dropin_points_timing.py --mode patched --steps 20" > $O/rocprof_c4.log 2>&1
stamp rocprof_pipeline
export OMP_NUM_THREADS=16
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_pipeline/stats -- python profiles/bench_pipeline.py > $O/pipeline.json 2> $O/pipeline.err
find $O/prof_pipeline -type f ! -name "*.csv" -delete
du -sh $O
stamp end

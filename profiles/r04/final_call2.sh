#!/bin/bash
# Round 4, closing evidence on the last product build (after final_call.sh: 24-bit record address, sample-major backward for K = 32,
# point-rasterizer set-up): python bench.py (the driver's command) + profiles/run_rocprof.sh.  The full test-suite log of the
# round is profiles/r04/final/tests_full.txt (final_call.sh); the suites of what changed since ran in the calls in between.
set -u
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
O=gpurun_out/r04final2
mkdir -p $O
stamp() { echo "== $1 $(date +%T)" | tee -a $O/steps.txt; }
stamp bench
timeout 600 python bench.py > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"
cut -c1-400 $O/bench.json
stamp rocprof
timeout 400 bash profiles/run_rocprof.sh $O/prof > $O/rocprof.log 2>&1
python profiles/summarize.py $O/prof $O/r04_final 1.0 > /dev/null 2>&1; cp profiles/traffic.json $O/traffic.json
grep -E "^\| mesh_fine|^\| mesh_backward|HBM traffic|VALU wave" $O/r04_final_rocprof.md | head -6
stamp tests
timeout 300 python -m pytest tests/test_gpu_bench_contract.py tests/test_gpu_cover.py tests/test_gpu_render_chain.py -q 2>&1 | tail -3
stamp end

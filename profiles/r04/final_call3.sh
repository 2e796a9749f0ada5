#!/bin/bash
# Round 4, closing evidence on the last product build (after final_call2.sh: short workspaces, binning with lane masks and
# bisected rectangles, four binning launches instead of six): python bench.py (the driver's command) + profiles/run_rocprof.sh
# + the contract / chain tests and the trimmed full-size config-3 test.
set -u
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
O=gpurun_out/r04final3
mkdir -p $O
stamp() { echo "== $1 $(date +%T)" | tee -a $O/steps.txt; }
stamp bench
timeout 600 python bench.py > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"
cut -c1-400 $O/bench.json
stamp rocprof
timeout 400 bash profiles/run_rocprof.sh $O/prof > $O/rocprof.log 2>&1
python profiles/summarize.py $O/prof $O/r04_final 1.0 > /dev/null 2>&1; cp profiles/traffic.json $O/traffic.json
grep -E "^\| mesh_fine|^\| mesh_backward|^\| bin_|HBM traffic|VALU wave" $O/r04_final_rocprof.md | head -12
stamp tests
timeout 500 python -m pytest tests/test_gpu_bench_contract.py tests/test_gpu_render_chain.py tests/test_gpu_baseline_sizes.py -q --durations=3 2>&1 | tail -8 | tee $O/tests.txt
stamp end

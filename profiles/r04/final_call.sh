#!/bin/bash
# Round 4 evidence call: (1) python bench.py (the driver's command, default steps; CPU baselines and drop-in legs included),
# (2) profiles/run_rocprof.sh: kernel trace + stats and the SQ / LDS / FETCH_SIZE / WRITE_SIZE passes of the same command, folded
# by profiles/summarize.py, (3) the whole `pytest -m gpu` suite on the shipped build.  Everything lands under gpurun_out/r04final/.
set -u
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
O=gpurun_out/r04final
mkdir -p $O
stamp() { echo "== $1 $(date +%T)" | tee -a $O/steps.txt; }
stamp start
rocminfo 2>/dev/null | grep -E "Marketing Name|Compute Unit|Max Clock" | tail -4 > $O/device.txt
stamp bench
timeout 600 python bench.py > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"
cut -c1-700 $O/bench.json
stamp rocprof
timeout 400 bash profiles/run_rocprof.sh $O/prof > $O/rocprof.log 2>&1
python profiles/summarize.py $O/prof $O/r04_final 1.0 > /dev/null 2>&1; cp profiles/traffic.json $O/traffic.json
grep -E "^\| mesh_fine|^\| mesh_backward|HBM traffic|VALU wave" $O/r04_final_rocprof.md | head -12
stamp tests
timeout 1500 python -m pytest tests -m gpu -q --durations=15 > $O/tests_full.txt 2>&1; echo "rc=$?" >> $O/tests_full.txt
tail -n 22 $O/tests_full.txt | cut -c1-200
stamp end

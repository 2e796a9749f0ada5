#!/bin/bash
# Round 3, closing GPU call (3.3 GPU-minutes left): the shipped build after the two binning changes (bin_fill offsets through
# LDS, row scan by shape).  Parity suites first; then `python bench.py` (the driver's command) with the rocprofv3 passes
# run WHILE its CPU-baseline legs occupy the host (the GPU is idle then; kernel durations are unaffected).
set -u
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
O=gpurun_out/r03final2
mkdir -p $O
stamp() { echo "== $1 $(date +%T)" | tee -a $O/steps.txt; }
stamp tests
timeout 120 python -m pytest tests/test_gpu_bench_launch_parity.py tests/test_gpu_cover.py tests/test_gpu_meshes.py tests/test_gpu_points_composite_interp.py \
  tests/test_gpu_reference_suite_replay.py tests/test_gpu_vs_reference_device_kernels.py -x -q -s > $O/tests.txt 2>&1; echo "rc=$?" >> $O/tests.txt
grep -E "^\[bench|passed|failed|rc=" $O/tests.txt | cut -c1-260 | tail -6
stamp bench
python bench.py > $O/bench.json 2> $O/bench.err &
BP=$!
sleep 28   # the timed region, the other configs, the scale sensitivity and the two drop-in subprocesses are over by then
stamp rocprof
timeout 120 bash profiles/run_rocprof.sh $O/prof > $O/rocprof.log 2>&1
stamp wait
wait $BP; echo "bench rc=$?"
cut -c1-500 $O/bench.json
stamp end

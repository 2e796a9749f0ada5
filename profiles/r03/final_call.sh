#!/bin/bash
# Round 3, last GPU call (14 GPU-minutes were left): the steps are ordered by what must not be lost, every one under its
# own timeout, everything written under gpurun_out/r03final/ as it is produced.
#   1. parity of the shipped build on the kernels the last commits touched (bench launch vs the reference's device
#      kernels, row cover, mesh suite)
#   2. `python bench.py` (the driver's command, default steps) -> bench.json
#   3. profiles/run_rocprof.sh (kernel trace + stats, SQ / LDS / FETCH_SIZE / WRITE_SIZE passes, each its own run)
#   4. soft-Phong pipeline record, then the point / shading / soft-Phong suites for as long as the call lasts
set -u
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
O=gpurun_out/r03final
mkdir -p $O
stamp() { echo "== $1 $(date +%T)" | tee -a $O/steps.txt; }
stamp start
rocminfo 2>/dev/null | grep -E "Marketing Name|Compute Unit|Max Clock" | tail -4 > $O/device.txt
stamp tests_a
timeout 260 python -m pytest tests/test_gpu_bench_launch_parity.py tests/test_gpu_cover.py tests/test_gpu_meshes.py -x -q -s --durations=12 \
  > $O/tests_a.txt 2>&1; echo "rc=$?" >> $O/tests_a.txt
grep -E "^\[|passed|failed|rc=|Error" $O/tests_a.txt | cut -c1-300 | tail -12
stamp bench
timeout 330 python bench.py > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"
cut -c1-900 $O/bench.json
stamp rocprof
timeout 300 bash profiles/run_rocprof.sh $O/prof > $O/rocprof.log 2>&1
tail -n 3 $O/rocprof.log
stamp pipeline
timeout 120 python profiles/bench_pipeline.py > $O/pipeline.json 2> $O/pipeline.err
cut -c1-600 $O/pipeline.json
stamp tests_b
timeout 600 python -m pytest tests/test_gpu_points_composite_interp.py tests/test_gpu_shading.py tests/test_gpu_soft_phong.py \
  tests/test_gpu_vs_reference_device_kernels.py tests/test_gpu_baseline_sizes.py tests/test_gpu_bench_contract.py -x -q --durations=12 \
  > $O/tests_b.txt 2>&1; echo "rc=$?" >> $O/tests_b.txt
tail -n 16 $O/tests_b.txt | cut -c1-200
stamp end

#!/bin/bash
# Round 4, GPU call 2: product = pixel masks + disjoint pairs + cubic reciprocal step + queues without payload for 17..64;
# base = the product of call 1.  exact_div brute force, A/B in one process, mesh suites, K sweeps.
set -u
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
O=gpurun_out/r04c2
mkdir -p $O
L=$PWD/pytorch3d_amd
stamp() { echo "== $1 $(date +%T)" | tee -a $O/steps.txt; }
stamp exact_div
timeout 120 ./profiles/microbench/exact_div_check.bin > $O/exact_div_check.txt 2>&1; echo "rc=$?" >> $O/exact_div_check.txt; cat $O/exact_div_check.txt
stamp exp
timeout 200 python profiles/exp_measure.py --torus-div 1.0 base=$L/libp3d_base.so > $O/exp_literal.jsonl 2> $O/exp_literal.txt; tail -n 3 $O/exp_literal.txt
timeout 200 python profiles/exp_measure.py --torus-div 1.5 base=$L/libp3d_base.so > $O/exp_light.jsonl 2> $O/exp_light.txt; tail -n 3 $O/exp_light.txt
stamp tests
timeout 600 python -m pytest tests/test_gpu_meshes.py tests/test_gpu_bench_launch_parity.py tests/test_gpu_cover.py tests/test_gpu_vs_reference_device_kernels.py \
  tests/test_gpu_reference_suite_replay.py -x -q --durations=8 > $O/tests.txt 2>&1
echo "rc=$?" >> $O/tests.txt; grep -E "passed|failed|rc=|Error|assert|^E " $O/tests.txt | cut -c1-300 | tail -15
stamp ksweep
timeout 200 python profiles/k_sweep.py 8 16 17 20 24 32 40 48 50 64 100 > $O/k_product.txt 2>&1; cat $O/k_product.txt
stamp end

#!/bin/bash
# Round 4, GPU call 3: no kernel spills VGPRs any more (build guard); K = 3, 5..7, 9..15 on pair queues with K live entries,
# 17..64 on queues without payload, points 65..99 on the 100-entry pair queue.  Overflow suites, A/B vs base, K sweeps.
set -u
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
O=gpurun_out/r04c3
mkdir -p $O
L=$PWD/pytorch3d_amd
stamp() { echo "== $1 $(date +%T)" | tee -a $O/steps.txt; }
stamp tests
timeout 900 python -m pytest tests/test_gpu_meshes.py tests/test_gpu_points_composite_interp.py tests/test_gpu_bench_launch_parity.py tests/test_gpu_cover.py \
  tests/test_gpu_vs_reference_device_kernels.py tests/test_gpu_reference_suite_replay.py tests/test_gpu_soft_phong.py tests/test_gpu_shading.py tests/test_gpu_baseline_sizes.py \
  -q --durations=8 > $O/tests.txt 2>&1
echo "rc=$?" >> $O/tests.txt; grep -E "passed|failed|rc=|Error|^E  |^FAILED" $O/tests.txt | cut -c1-300 | tail -25
stamp exp
timeout 200 python profiles/exp_measure.py --torus-div 1.0 base=$L/libp3d_base.so > $O/exp_literal.jsonl 2> $O/exp_literal.txt; tail -n 3 $O/exp_literal.txt
stamp ksweep
timeout 200 python profiles/k_sweep.py 3 4 5 6 8 9 12 15 16 17 24 32 40 48 49 64 100 > $O/k_product.txt 2>&1; cat $O/k_product.txt
ABL_BATCH=64 timeout 200 python profiles/k_sweep.py 4 12 16 32 > $O/k_product_b64.txt 2>&1; cat $O/k_product_b64.txt
stamp points
timeout 100 python profiles/points_k_sweep.py 8 10 16 32 50 64 80 100 > $O/points_k.txt 2>&1; tail -n 12 $O/points_k.txt
stamp end

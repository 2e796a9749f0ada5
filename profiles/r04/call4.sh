#!/bin/bash
# Round 4, GPU call 4: the committed product (no spilled VGPRs anywhere, SCC clobbers declared) through the suites the round
# touched; K sweeps product vs the long16 variant (9..16 on a queue without payload at four waves per SIMD).
set -u
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
O=gpurun_out/r04c4
mkdir -p $O
L=$PWD/pytorch3d_amd
stamp() { echo "== $1 $(date +%T)" | tee -a $O/steps.txt; }
stamp tests
timeout 900 python -m pytest tests/test_gpu_meshes.py tests/test_gpu_points_composite_interp.py tests/test_gpu_bench_launch_parity.py tests/test_gpu_cover.py \
  tests/test_gpu_vs_reference_device_kernels.py tests/test_gpu_reference_suite_replay.py tests/test_gpu_soft_phong.py tests/test_gpu_shading.py \
  -q --durations=8 -k "not interp_properties_at_full" > $O/tests.txt 2>&1
echo "rc=$?" >> $O/tests.txt; grep -E "passed|failed|rc=|Error|^E  |^FAILED" $O/tests.txt | cut -c1-300 | tail -25
stamp ksweep
timeout 200 python profiles/k_sweep.py 9 12 15 16 > $O/k_product.txt 2>&1; cat $O/k_product.txt
P3D_LIB_PATH=$L/libp3d_long16.so timeout 200 python profiles/k_sweep.py 9 12 15 16 > $O/k_long16.txt 2>&1; cat $O/k_long16.txt
ABL_BATCH=64 timeout 200 python profiles/k_sweep.py 12 16 > $O/k_product_b64.txt 2>&1; cat $O/k_product_b64.txt
ABL_BATCH=64 P3D_LIB_PATH=$L/libp3d_long16.so timeout 200 python profiles/k_sweep.py 12 16 > $O/k_long16_b64.txt 2>&1; cat $O/k_long16_b64.txt
P3D_LIB_PATH=$L/libp3d_long16.so timeout 100 python profiles/r04/k_parity_probe.py 9 12 13 16 2>&1 | cut -c1-100 | grep -v "mismatches 0 floats \[0, 0, 0\]"
stamp points
timeout 100 python profiles/points_k_sweep.py 10 80 > $O/points_k.txt 2>&1; tail -n 3 $O/points_k.txt
stamp end

#!/bin/bash
# Round 3, GPU call 1: VALU issue calibration, the prepared build-flag variants (measured in one process against the
# product library), the point-queue variants, the two new contract / parity tests.
set -u
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
O=gpurun_out/r03
mkdir -p $O
rocminfo | grep -E "Marketing Name|Compute Unit|Max Clock" | head -6 > $O/device.txt 2>&1
echo "== microbench"; date +%T
timeout 300 ./profiles/microbench/valu_issue.bin > $O/valu_issue_mi355x.txt 2>&1; tail -n 60 $O/valu_issue_mi355x.txt
echo "== variants"; date +%T
L=pytorch3d_amd
timeout 600 python profiles/exp_measure.py pairs=$L/libp3d_pairs.so packed=$L/libp3d_packed.so both=$L/libp3d_both.so key64=$L/libp3d_key64.so \
  bwdpk=$L/libp3d_bwdpk.so all=$L/libp3d_all.so fill16=$L/libp3d_fill16.so fill0=$L/libp3d_fill0.so > $O/exp_measure.jsonl 2> $O/exp_measure.txt
tail -n 14 $O/exp_measure.txt
echo "== points"; date +%T
timeout 200 python profiles/points_k_sweep.py 8 10 16 32 40 50 64 100 > $O/points_product.txt 2>&1
for v in ppairs pkey64; do
  P3D_LIB_PATH=$PWD/$L/libp3d_$v.so timeout 200 python profiles/points_k_sweep.py 8 10 16 32 40 50 64 100 > $O/points_$v.txt 2>&1
  P3D_LIB_PATH=$PWD/$L/libp3d_$v.so timeout 300 python -m pytest tests/test_gpu_points_composite_interp.py -x -q > $O/tests_$v.txt 2>&1
  tail -n 2 $O/tests_$v.txt
done
tail -n 9 $O/points_product.txt $O/points_ppairs.txt $O/points_pkey64.txt
echo "== new tests"; date +%T
timeout 600 python -m pytest tests/test_gpu_bench_launch_parity.py tests/test_gpu_bench_contract.py -x -q -s > $O/tests_new.txt 2>&1
tail -n 15 $O/tests_new.txt
date +%T

#!/bin/bash
# round 5, call 17: SQ counters of point_tile_sorted_kernel and of the register-queue kernel at K = 10 (1M points)
set -u
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp OMP_NUM_THREADS=16
O=gpurun_out/r05c17
mkdir -p $O
for v in amd r5queues; do
  P3D_LIB_PATH=$PWD/pytorch3d_amd/libp3d_$v.so rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_LDS \
    --kernel-trace --output-format csv -d $O/pmc_$v -- python profiles/points_k_sweep.py 10 > $O/pmc_$v.log 2>&1
  P3D_LIB_PATH=$PWD/pytorch3d_amd/libp3d_$v.so rocprofv3 --pmc SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS SQ_INSTS_SMEM SQ_INST_CYCLES_VMEM \
    --kernel-trace --output-format csv -d $O/pmc2_$v -- python profiles/points_k_sweep.py 10 > $O/pmc2_$v.log 2>&1
done
find $O -type f ! -name "*.csv" ! -name "*.log" -delete

#!/bin/bash
# round 5, call 10: the wave-decoupled fine stage for meshes (experiment library libp3d_wavetiles.so, -DP3D_EXP_WAVE_TILES) against the
# product in ONE process on the bench launch: kernel times, bit parity, then the SQ counters of both kernels
set -u
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp OMP_NUM_THREADS=16
O=gpurun_out/r05c10
mkdir -p $O
L=$PWD/pytorch3d_amd
timeout 300 python profiles/exp_measure.py wavetiles=$L/libp3d_wavetiles.so > $O/exp_measure.jsonl 2> $O/exp_measure.txt; tail -5 $O/exp_measure.txt
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_LDS \
  --kernel-trace --output-format csv -d $O/pmc_sq -- python profiles/exp_measure.py --iters 10 wavetiles=$L/libp3d_wavetiles.so > $O/pmc.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- python profiles/exp_measure.py --iters 10 wavetiles=$L/libp3d_wavetiles.so > $O/stats.log 2>&1
find $O -type f ! -name "*.csv" ! -name "*.txt" ! -name "*.jsonl" ! -name "*.log" -delete
P3D_LIB_PATH=$L/libp3d_wavetiles.so timeout 200 python -m pytest tests/test_gpu_bench_launch_parity.py tests/test_gpu_cover.py -x -q -p no:cacheprovider 2>&1 | tail -3

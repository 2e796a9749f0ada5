#!/bin/bash
# Round 3, GPU call 10: the round's closing records on the shipped build -- K sweeps (meshes: 8 meshes and the full batch;
# points), the section 8(f) kernels (bench_configs.py), the single-image latency (c2_latency.py).
set -u
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
O=gpurun_out/r03c10
mkdir -p $O
timeout 120 python profiles/k_sweep.py 1 2 4 8 12 16 32 50 100 2>&1 | grep K= | tee $O/mesh_k_final_8meshes.txt
ABL_BATCH=64 timeout 120 python profiles/k_sweep.py 4 8 16 2>&1 | grep K= | tee $O/mesh_k_final_batch64.txt
timeout 120 python profiles/points_k_sweep.py 1 8 10 16 32 50 64 100 2>&1 | grep K= | tee $O/points_k_final.txt
timeout 60 python profiles/c2_latency.py 2>&1 | tail -n 12 | tee $O/config2_latency.txt
timeout 200 python profiles/bench_configs.py > $O/configs.json 2> $O/configs.err; cut -c1-400 $O/configs.json

#!/bin/bash
# round 6, call 31: product (table read in front of the staging branch) vs the read guarded by "the wave stages a face" vs the read inside the branch (before call 28)
set -u
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
O=gpurun_out/r06c31
mkdir -p $O
L=$PWD/pytorch3d_amd
for i in 1 2 3; do
timeout 400 python profiles/exp_measure.py --iters 100 guard=$L/libp3d_guard.so nohoist=$L/libp3d_nohoist.so > $O/exp_$i.jsonl 2> $O/exp_$i.txt; tail -n 4 $O/exp_$i.txt
done

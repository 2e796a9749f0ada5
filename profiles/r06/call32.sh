#!/bin/bash
# round 6, call 32: is there an order effect in exp_measure?  the product library again as a variant, in both positions
set -u
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
O=gpurun_out/r06c32
mkdir -p $O
L=$PWD/pytorch3d_amd
for i in 1 2; do
timeout 400 python profiles/exp_measure.py --iters 100 nohoist=$L/libp3d_nohoist.so same=$L/libp3d_same.so > $O/a_$i.jsonl 2> $O/a_$i.txt; tail -n 4 $O/a_$i.txt
timeout 400 python profiles/exp_measure.py --iters 100 same=$L/libp3d_same.so nohoist=$L/libp3d_nohoist.so > $O/b_$i.jsonl 2> $O/b_$i.txt; tail -n 4 $O/b_$i.txt
done

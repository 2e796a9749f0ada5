#!/bin/bash
# round 6, call 4: A/B of the two-step-segment insertion network (variant "seg") against the product, in one process
set -u
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
O=gpurun_out/r06c4
mkdir -p $O
L=$PWD/pytorch3d_amd
SPEC=""
for v in "$@"; do SPEC="$SPEC $v=$L/libp3d_$v.so"; done
timeout 300 python profiles/exp_measure.py --iters 100 $SPEC > $O/exp_measure.jsonl 2> $O/exp_measure.txt; tail -n $(( $# + 2 )) $O/exp_measure.txt

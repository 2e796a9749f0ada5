#!/bin/bash
# round 6, call 2: insertion-position histogram probe (variant library)
set -u
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
O=gpurun_out/r06c2
mkdir -p $O
P3D_LIB_PATH=$PWD/pytorch3d_amd/libp3d_probe.so timeout 300 python profiles/r06/probe_depth.py > $O/probe_insert.txt 2>&1
cat $O/probe_insert.txt

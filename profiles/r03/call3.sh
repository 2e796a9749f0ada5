#!/bin/bash
set -u
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
O=gpurun_out/r03
mkdir -p $O
L=$PWD/pytorch3d_amd
echo "product"; ABL_BATCH=64 timeout 300 python profiles/k_sweep.py 4 8 16 2>&1 | grep K= | tee $O/mesh_k_product.txt
for v in pairs key64; do
echo $v; P3D_LIB_PATH=$L/libp3d_$v.so ABL_BATCH=64 timeout 300 python profiles/k_sweep.py 4 8 16 2>&1 | grep K= | tee $O/mesh_k_$v.txt
done
# non-PC flags (persp only / none) use the plain instantiation: does the pair queue help there too?

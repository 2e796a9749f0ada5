#!/bin/bash
# round 6, call 1: depth pre-test probe (variant library) + the product's bench line as the round's starting point
set -u
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
O=gpurun_out/r06c1
mkdir -p $O
P3D_LIB_PATH=$PWD/pytorch3d_amd/libp3d_probe.so timeout 300 python profiles/r06/probe_depth.py > $O/probe_depth.txt 2>&1
cat $O/probe_depth.txt
timeout 200 python bench.py --steps 100 --no-cpu-baseline --no-dropin > $O/bench_product.json 2> $O/bench_product.err
python -c "
import json;b=json.load(open('$O/bench_product.json'));print(round(b['value'],1), 'Mpix/s', round(b['ms_per_step'],4), 'ms', b['kernels_ms'])"

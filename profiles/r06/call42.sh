#!/bin/bash
# round 6, call 42: mesh_backward (per-face records form) at seven waves per SIMD (100-slot tables) vs six (116): bench alternating, then parity
set -u
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
O=gpurun_out/r06c42
mkdir -p $O
L=$PWD/pytorch3d_amd
for i in 1 2 3; do
for v in amd occ7; do
P3D_LIB_PATH=$L/libp3d_$v.so timeout 600 python bench.py --steps 100 --warmup 5 --no-cpu-baseline --no-dropin --no-reference-device --no-other-configs > $O/bench_${v}_$i.json 2> $O/bench.err
python - <<PY
import json
b=json.loads([l for l in open('$O/bench_${v}_$i.json') if l.startswith('{')][0])
print('$v', round(b['value'],1),'Mpix/s',round(b['ms_per_step'],4),'ms', 'fine', b['kernels_ms']['mesh_fine'], 'bwd', b['kernels_ms']['mesh_backward'], 'light', b.get('workload_torus_div_1.5',{}).get('kernels_ms',{}).get('mesh_backward'))
PY
done
done
P3D_LIB_PATH=$L/libp3d_occ7.so timeout 600 python -m pytest tests/test_gpu_meshes.py tests/test_gpu_cover.py tests/test_gpu_bench_launch_parity.py tests/test_gpu_baseline_sizes.py -x -q -m gpu 2>&1 | tail -n 1

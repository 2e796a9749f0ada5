#!/bin/bash
# round 5, call 35: the depths of the list entries beside the ids (bin_fill writes them, the tile-sorted kernel's depth sort reads
# them with the list) against gathering them (-DP3D_NO_LISTZ): config 4 through the drop-in, K sweep; the point suites on the product
set -u
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp OMP_NUM_THREADS=16
O=gpurun_out/r05c35
mkdir -p $O
V=$PWD/pytorch3d_amd/libp3d_nolistz.so
for rep in 1 2; do
for lib in product nolistz; do
  if [ $lib = nolistz ]; then export P3D_LIB_PATH=$V; else unset P3D_LIB_PATH; fi
  timeout 200 python profiles/dropin_points_timing.py --mode patched --steps 50 > $O/points_${lib}_$rep.json 2>&1
  python - <<PY
import json
j=json.loads([l for l in open("$O/points_${lib}_$rep.json") if l.startswith("{")][-1])
k=j["our_kernels_ms_per_step"]
print("$lib", round(j["ms_per_step"],4), "points_fine", k["points_fine"], "bin_fill", k["bin_fill"], "sum", j["our_kernels_sum_ms"])
PY
done
done
unset P3D_LIB_PATH
echo "[product]"; timeout 200 python profiles/points_k_sweep.py 1 8 10 16 24 28 32 > $O/k_sweep_product.txt 2>&1; grep K= $O/k_sweep_product.txt
echo "[nolistz]"; P3D_LIB_PATH=$V timeout 200 python profiles/points_k_sweep.py 1 8 10 16 24 28 32 > $O/k_sweep_nolistz.txt 2>&1; grep K= $O/k_sweep_nolistz.txt
timeout 600 python -m pytest tests/test_gpu_points_composite_interp.py tests/test_gpu_points_renderer_dropin.py tests/test_gpu_short_workspace.py tests/test_gpu_vs_reference_device_kernels.py tests/test_gpu_baseline_sizes.py -x -q -p no:cacheprovider > $O/tests.txt 2>&1; tail -3 $O/tests.txt

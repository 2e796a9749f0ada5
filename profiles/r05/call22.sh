#!/bin/bash
# round 5, call 22: the adjacent-lane table flush as the product (vertex ids of kCorners loaded up front) against the flush of
# rounds 2-4 (-DP3D_FLUSH_LANE_PER_ROW): mesh_backward on both bench batches and K = 4 / 16, the config-4 chain, the soft-Phong
# pipeline; the whole GPU suite on the product.
set -u
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp OMP_NUM_THREADS=16
O=gpurun_out/r05c22
mkdir -p $O
V=$PWD/pytorch3d_amd/libp3d_rowflush.so
for td in 1.0 1.5; do
  timeout 300 python profiles/exp_measure.py --iters 40 --torus-div $td rowflush=$V > $O/exp_measure_$td.json 2> $O/exp_measure_$td.txt; tail -3 $O/exp_measure_$td.txt
done
for k in 4 16; do
  timeout 300 python profiles/exp_measure.py --iters 20 --batch 16 --faces-per-pixel $k rowflush=$V > $O/exp_measure_k$k.json 2> $O/exp_measure_k$k.txt; tail -3 $O/exp_measure_k$k.txt
done
for lib in product rowflush; do
  if [ $lib = rowflush ]; then export P3D_LIB_PATH=$V; else unset P3D_LIB_PATH; fi
  timeout 200 python profiles/dropin_points_timing.py --mode patched --steps 50 > $O/points_$lib.json 2>&1
  python - <<PY
import json
j=json.loads([l for l in open("$O/points_$lib.json") if l.startswith("{")][-1])
print("$lib", round(j["ms_per_step"],4), j["our_kernels_ms_per_step"])
PY
  timeout 300 python profiles/bench_pipeline.py > $O/pipeline_$lib.json 2> $O/pipeline_$lib.err
  python - <<PY
import json
try:
    j=json.loads([l for l in open("$O/pipeline_$lib.json") if l.startswith("{")][-1])
    for k in ("phong_then_blend","fused_soft_phong"): print("$lib", k, j[k]["wall_ms"], j[k]["kernels_ms"])
except Exception as e: print("pipeline $lib", e)
PY
done
unset P3D_LIB_PATH
( time timeout 900 python -m pytest tests/ -x -q -m gpu -p no:cacheprovider ) > $O/tests.txt 2>&1; tail -5 $O/tests.txt

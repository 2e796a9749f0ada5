#!/bin/bash
# round 5, call 21: (a) microbenchmark of global float atomics -- a lane per row against the values of a row in adjacent lanes;
# (b) the table flush of wave_table.h with adjacent lanes (-DP3D_FLUSH_ADJ) against the product: mesh_backward on the bench batch,
# the config-4 chain, the soft-Phong pipeline; the whole GPU suite on the variant.
set -u
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp OMP_NUM_THREADS=16
O=gpurun_out/r05c21
mkdir -p $O
V=$PWD/pytorch3d_amd/libp3d_fadj.so
timeout 120 ./profiles/microbench/global_atomic.bin > $O/global_atomic.txt 2>&1; cat $O/global_atomic.txt
timeout 300 python profiles/exp_measure.py --iters 40 fadj=$V > $O/exp_measure.json 2> $O/exp_measure.txt; tail -6 $O/exp_measure.txt
for lib in product fadj; do
  if [ $lib = fadj ]; then export P3D_LIB_PATH=$V; else unset P3D_LIB_PATH; fi
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
    print("$lib pipeline", json.dumps(j)[:900])
except Exception as e: print("pipeline $lib", e)
PY
done
export P3D_LIB_PATH=$V
( time timeout 900 python -m pytest tests/ -x -q -m gpu -p no:cacheprovider ) > $O/tests_fadj.txt 2>&1; tail -5 $O/tests_fadj.txt

#!/bin/bash
# round 5, call 26: the cheap CUDA tie order (TIES instantiations mark the pixels, the replay visits only those): cost against the
# plain forward on config 3, the tie-order tests against the reference's device kernels; the single-block offsets scan and the
# unrolled row scan of the binning on config 4.
set -u
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp OMP_NUM_THREADS=16
O=gpurun_out/r05c26
mkdir -p $O
timeout 300 python profiles/tie_order_timing.py 8 1 4 16 > $O/tie_order_timing.txt 2>&1; cat $O/tie_order_timing.txt | tail -6
timeout 600 python -m pytest tests/test_gpu_vs_reference_device_kernels.py tests/test_gpu_meshes.py -x -q -p no:cacheprovider > $O/tests_mesh.txt 2>&1; tail -3 $O/tests_mesh.txt
timeout 200 python profiles/dropin_points_timing.py --mode patched --steps 50 > $O/points.json 2>&1
python - <<PY
import json
j=json.loads([l for l in open("$O/points.json") if l.startswith("{")][-1])
print(round(j["ms_per_step"],4), j["our_kernels_ms_per_step"], j["our_kernels_sum_ms"])
PY
timeout 400 python -m pytest tests/test_gpu_points_composite_interp.py tests/test_gpu_points_renderer_dropin.py tests/test_gpu_short_workspace.py -x -q -p no:cacheprovider > $O/tests_points.txt 2>&1; tail -3 $O/tests_points.txt

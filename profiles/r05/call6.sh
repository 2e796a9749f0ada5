#!/bin/bash
# round 5, call 6: ADVICE fixes + the speculative un-clipped path of the patched MeshRasterizer.forward + OMP-sized drop-in children
set -u
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
O=gpurun_out/r05c6
mkdir -p $O
stamp() { echo "== $1 $(date +%T)" | tee -a $O/steps.txt; }
stamp tests
timeout 900 python -m pytest tests/test_gpu_blending.py tests/test_gpu_points_composite_interp.py tests/test_gpu_cover.py tests/test_gpu_points_renderer_dropin.py \
  tests/test_gpu_reference_own_tests.py tests/test_gpu_short_workspace.py tests/test_gpu_render_chain.py -x -q -p no:cacheprovider > $O/tests.txt 2>&1
echo "tests rc=$?"; tail -4 $O/tests.txt
stamp dropin
export OMP_NUM_THREADS=16
for m in c_only patched; do
  timeout 200 python profiles/dropin_timing.py --mode $m > $O/dropin_mesh_$m.json 2>$O/dropin_mesh_$m.err
  python -c "import json;j=json.load(open('$O/dropin_mesh_$m.json'));print('mesh $m', round(j['ms_per_step'],4), j['our_kernels_sum_ms'], j.get('patched_calls',{}).get('MeshRasterizer.forward: no vertex behind z_clip, un-clipped fused path kept'))"
  timeout 120 python profiles/dropin_points_timing.py --mode $m > $O/dropin_points_$m.json 2>/dev/null
  python -c "import json;j=json.load(open('$O/dropin_points_$m.json'));print('points $m', round(j['ms_per_step'],4), j['our_kernels_sum_ms'])"
done
stamp end

#!/bin/bash
# round 5, call 8: K > 48 two-pass scheme (first pass: 32 or 48 register entries, private-memory kernel redoes the tiles with a full
# queue) vs round 4's private-memory kernel for every tile; the patched drop-in after the isempty() sync was removed
set -u
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp OMP_NUM_THREADS=16
O=gpurun_out/r05c8
mkdir -p $O
stamp() { echo "== $1 $(date +%T)" | tee -a $O/steps.txt; }
stamp tests
timeout 600 python -m pytest tests/test_gpu_meshes.py tests/test_gpu_cover.py tests/test_gpu_vs_reference_device_kernels.py tests/test_gpu_short_workspace.py -x -q -p no:cacheprovider > $O/tests.txt 2>&1
echo "tests rc=$?"; tail -3 $O/tests.txt
stamp sweep
for v in amd redo48 r5base; do
  P3D_LIB_PATH=$PWD/pytorch3d_amd/libp3d_$v.so timeout 200 python profiles/k_sweep.py 48 49 64 100 150 > $O/k_sweep_$v.txt 2>&1; echo "[$v]"; grep "K=" $O/k_sweep_$v.txt
done
stamp dropin
timeout 200 python profiles/dropin_timing.py --mode patched --steps 50 > $O/dropin_mesh_patched.json 2>/dev/null
python -c "import json;j=json.load(open('$O/dropin_mesh_patched.json'));print('mesh patched', round(j['ms_per_step'],4), j['our_kernels_sum_ms'])"
stamp end

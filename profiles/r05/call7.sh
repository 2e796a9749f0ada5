#!/bin/bash
# round 5, call 7: where the 0.46 ms between the patched drop-in's wall and its kernels go (host profile), reference suite re-check
set -u
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp OMP_NUM_THREADS=16
O=gpurun_out/r05c7
mkdir -p $O
timeout 200 python profiles/dropin_timing.py --mode patched --steps 50 --cprofile > $O/dropin_mesh_patched.json 2> $O/cprofile_patched.txt
python -c "import json;j=json.load(open('$O/dropin_mesh_patched.json'));print('mesh patched', round(j['ms_per_step'],4), j['our_kernels_sum_ms'])"
timeout 400 python -m pytest tests/test_gpu_reference_own_tests.py -x -q -p no:cacheprovider 2>&1 | tail -3

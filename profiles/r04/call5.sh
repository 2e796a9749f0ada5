#!/bin/bash
# Round 4, GPU call 5: the drop-in path after the host-side work (lean Meshes.offset_verts, cached camera matrices, the row cover
# found by the reference-style backward call); the reference's own suite in both shim modes.
set -u
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
O=gpurun_out/r04c5
mkdir -p $O
stamp() { echo "== $1 $(date +%T)" | tee -a $O/steps.txt; }
stamp dropin
for div in 1.0 1.5; do for mode in c_only patched; do
  timeout 200 python profiles/dropin_timing.py --mode $mode --torus-div $div --steps 20 > $O/dropin_${mode}_$div.json 2> $O/dropin_${mode}_$div.err
  python -c "
import json;b=json.load(open('$O/dropin_${mode}_$div.json'));print('$mode', '$div', round(b['ms_per_step'],3),'ms; ours', b['our_kernels_sum_ms'], b['our_kernels_ms_per_step'].get('mesh_backward'), b['cover_recalls_hit_miss'], b.get('patched_calls'))"
done; done
stamp cover
timeout 300 python -m pytest tests/test_gpu_cover.py -q -x 2>&1 | tail -3
stamp refsuite
timeout 1500 python -m pytest tests/test_gpu_reference_own_tests.py -q -x -s 2>&1 | tail -15 | cut -c1-300
stamp end

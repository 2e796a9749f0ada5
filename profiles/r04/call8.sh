#!/bin/bash
# Round 4, GPU call 8: short workspaces of rasterize_meshes (device-side overflow flag + naive fallback): parity tests, the
# binning / mesh / point suites on the re-ordered arena, and the cost on the bench batch: HEAD's library (base), this build with
# the worst-case workspace (product), with a short one that fits (short: + one empty launch), with one that does not (overflow:
# the naive kernel writes the batch).
set -u
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
O=gpurun_out/r04c8
mkdir -p $O
L=$PWD/pytorch3d_amd
stamp() { echo "== $1 $(date +%T)" | tee -a $O/steps.txt; }
stamp tests_short
timeout 300 python -m pytest tests/test_gpu_short_workspace.py -q -x 2>&1 | tail -5 | tee $O/tests_short.txt
stamp measure
timeout 240 python profiles/exp_measure.py --iters 40 base=$L/libp3d_base.so short=$L/libp3d_amd.so@0 overflow=$L/libp3d_amd.so@1 \
  > $O/measure.json 2> $O/measure.txt; tail -6 $O/measure.txt
stamp tests_bins
timeout 420 python -m pytest tests/test_gpu_meshes.py tests/test_gpu_points_composite_interp.py tests/test_gpu_cover.py -q -x 2>&1 | tail -4 | tee $O/tests_bins.txt
stamp end

#!/bin/bash
# Round 4, GPU call 9: bin_count / bin_fill with row and column lane masks (a lane per bin of the wave's union, a lane's own
# rectangle in the placement pass) and bisected bin rectangles -- against the library of call 8 (base) on the bench batch, both
# workloads, and the suites that pin the bin lists (mesh / point coarse operators vs the oracle, short workspaces, cover).
set -u
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
O=gpurun_out/r04c9
mkdir -p $O
L=$PWD/pytorch3d_amd
stamp() { echo "== $1 $(date +%T)" | tee -a $O/steps.txt; }
stamp measure
timeout 200 python profiles/exp_measure.py --iters 40 base=$L/libp3d_base.so > $O/measure.json 2> $O/measure.txt; tail -4 $O/measure.txt
timeout 200 python profiles/exp_measure.py --iters 40 --torus-div 1.5 base=$L/libp3d_base.so > $O/measure_light.json 2> $O/measure_light.txt; tail -3 $O/measure_light.txt
stamp tests_bins
timeout 500 python -m pytest tests/test_gpu_meshes.py tests/test_gpu_points_composite_interp.py tests/test_gpu_cover.py tests/test_gpu_short_workspace.py -q -x 2>&1 | tail -4 | tee $O/tests_bins.txt
stamp end

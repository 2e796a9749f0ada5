#!/bin/bash
# Round 4, GPU call 11: two launches fewer in the binning of batches of <= 64 elements (no plan kernel: the count pass
# publishes the chunk table; row scan + block sums in one kernel) against the library of call 9's commit (base).
set -u
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
O=gpurun_out/r04c11
mkdir -p $O
L=$PWD/pytorch3d_amd
timeout 200 python profiles/exp_measure.py --iters 60 base=$L/libp3d_base.so > $O/measure.json 2> $O/measure.txt; tail -3 $O/measure.txt
timeout 200 python profiles/exp_measure.py --iters 60 --torus-div 1.5 base=$L/libp3d_base.so > $O/measure_light.json 2> $O/measure_light.txt; tail -3 $O/measure_light.txt
timeout 500 python -m pytest tests/test_gpu_meshes.py tests/test_gpu_points_composite_interp.py tests/test_gpu_cover.py tests/test_gpu_short_workspace.py -q -x 2>&1 | tail -3 | tee $O/tests_bins.txt

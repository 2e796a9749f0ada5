#!/bin/bash
# Round 3, GPU call 8: the depth bound (libp3d_filter.so) and the bound + fast / slow interleaved queue moves (the product
# build of this tree) against the build of the previous commit (libp3d_base.so), in one process with bit parity; the mixed
# entry-step shapes in the class microbenchmark; the mesh parity suites on the new product build.
set -u
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
O=gpurun_out/r03c8
mkdir -p $O
L=$PWD/pytorch3d_amd
timeout 60 ./profiles/microbench/valu_classes.bin > $O/valu_classes_mi355x.txt 2>&1; tail -n 8 $O/valu_classes_mi355x.txt | cut -c1-160
timeout 200 python profiles/exp_measure.py base=$L/libp3d_base.so filter=$L/libp3d_filter.so > $O/exp_measure.jsonl 2> $O/exp_measure.txt; tail -n 5 $O/exp_measure.txt
timeout 200 python -m pytest tests/test_gpu_bench_launch_parity.py tests/test_gpu_cover.py tests/test_gpu_meshes.py tests/test_gpu_vs_reference_device_kernels.py tests/test_gpu_reference_suite_replay.py -x -q > $O/tests.txt 2>&1; tail -n 3 $O/tests.txt
timeout 100 python bench.py --steps 100 --no-cpu-baseline --no-other-configs --no-dropin > $O/bench_short.json 2>/dev/null; cut -c1-330 $O/bench_short.json; python -c "
import json;b=json.load(open('$O/bench_short.json'));print(b['kernels_ms'])"

#!/bin/bash
# A/B call on the GPU box: variants built by profiles/ab_variant.py against the product library.
#   gpurun --timeout 400 -- 'bash profiles/ab_call.sh NAME [NAME ...]'
# 1. exp_measure.py: all libraries in one process on the bench launch (kernel times, bit parity of the forward against the
#    product, gradient deviation);  2. per variant: the mesh / point / bin parity suites and a short bench run ON that library.
set -u
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
O=gpurun_out/ab
mkdir -p $O
L=$PWD/pytorch3d_amd
SPEC=""
for v in "$@"; do SPEC="$SPEC $v=$L/libp3d_$v.so"; done
timeout 240 python profiles/exp_measure.py $SPEC > $O/exp_measure.jsonl 2> $O/exp_measure.txt; tail -n $(( $# + 2 )) $O/exp_measure.txt
for v in "$@"; do
  P3D_LIB_PATH=$L/libp3d_$v.so timeout 240 python -m pytest tests/test_gpu_bench_launch_parity.py tests/test_gpu_cover.py tests/test_gpu_meshes.py \
    tests/test_gpu_points_composite_interp.py tests/test_gpu_reference_suite_replay.py tests/test_gpu_vs_reference_device_kernels.py -x -q > $O/tests_$v.txt 2>&1
  echo "[$v] $(tail -n 1 $O/tests_$v.txt)"
  P3D_LIB_PATH=$L/libp3d_$v.so timeout 100 python bench.py --steps 100 --no-cpu-baseline --no-dropin --no-reference-device > $O/bench_$v.json 2>/dev/null
  python -c "
import json;b=json.load(open('$O/bench_$v.json'));print('[$v]', round(b['value'],1), 'Mpix/s', round(b['ms_per_step'],4), 'ms', b['kernels_ms']); print('[$v]', {k:(x['wall_ms'],x['kernels_ms']) for k,x in b['other_configs'].items()})"
done

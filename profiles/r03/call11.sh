#!/bin/bash
# Round 3, GPU call 11: bin_fill with the list offsets staged through LDS (libp3d_binfill.so) against the product, then the
# suites that pin the bin lists on that library.
set -u
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
O=gpurun_out/r03c11
mkdir -p $O
L=$PWD/pytorch3d_amd
timeout 200 python profiles/exp_measure.py binfill=$L/libp3d_binfill.so > $O/exp_measure.jsonl 2> $O/exp_measure.txt; tail -n 4 $O/exp_measure.txt
python - <<PY
import json
for l in open("$O/exp_measure.jsonl"):
    d=json.loads(l); print(d["variant"], d["ms_per_step"], {k:v for k,v in d["kernels_ms"].items() if k.startswith("bin") or k.startswith("gather")})
PY
P3D_LIB_PATH=$L/libp3d_binfill.so timeout 300 python -m pytest tests/test_gpu_meshes.py tests/test_gpu_points_composite_interp.py tests/test_gpu_reference_suite_replay.py tests/test_gpu_cover.py tests/test_gpu_bench_launch_parity.py -x -q > $O/tests.txt 2>&1; tail -n 3 $O/tests.txt
P3D_LIB_PATH=$L/libp3d_binfill.so timeout 100 python bench.py --steps 100 --no-cpu-baseline --no-dropin > $O/bench_short.json 2>/dev/null; python -c "
import json;b=json.load(open('$O/bench_short.json'));print(b['value'], b['ms_per_step'], b['kernels_ms']); print({k:v['kernels_ms'] for k,v in b['other_configs'].items()})"

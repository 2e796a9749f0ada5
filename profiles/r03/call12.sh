#!/bin/bash
# Round 3, GPU call 12: binning variants against the product in one process -- 512-primitive chunks (chunk512), the row
# scan with a thread per row (scanrows), both (c512s); then bench kernels + the bin-pinning suites on each candidate.
set -u
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
O=gpurun_out/r03c12
mkdir -p $O
L=$PWD/pytorch3d_amd
timeout 200 python profiles/exp_measure.py chunk512=$L/libp3d_chunk512.so scanrows=$L/libp3d_scanrows.so c512s=$L/libp3d_c512s.so > $O/exp_measure.jsonl 2> $O/exp_measure.txt; tail -n 6 $O/exp_measure.txt
python - <<PY
import json
for l in open("$O/exp_measure.jsonl"):
    d=json.loads(l); k=d["kernels_ms"]; print(d["variant"], round(d["ms_per_step"],4), {a:b for a,b in k.items() if a.startswith("bin")}, "bin sum", round(sum(b for a,b in k.items() if a.startswith("bin")),4))
PY
for v in scanrows c512s; do
P3D_LIB_PATH=$L/libp3d_$v.so timeout 200 python -m pytest tests/test_gpu_meshes.py tests/test_gpu_points_composite_interp.py tests/test_gpu_reference_suite_replay.py tests/test_gpu_bench_launch_parity.py -x -q > $O/tests_$v.txt 2>&1; tail -n 2 $O/tests_$v.txt
P3D_LIB_PATH=$L/libp3d_$v.so timeout 100 python bench.py --steps 100 --no-cpu-baseline --no-dropin > $O/bench_$v.json 2>/dev/null; python -c "
import json;b=json.load(open('$O/bench_$v.json'));print('$v', b['value'], b['ms_per_step'], b['kernels_ms']); print({k:(v['wall_ms'],v['kernels_ms']) for k,v in b['other_configs'].items()})"
done

#!/bin/bash
# round 6, call 8: the bench contract tests (8 ranks over gloo, uneven deal, dry run), the CUDA-order and compositor suites on the
# rebuilt library, then the default `python bench.py` (the driver's command) with every leg
set -u
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
O=gpurun_out/r06c8
mkdir -p $O
( time timeout 900 python -m pytest tests/test_gpu_bench_contract.py tests/test_gpu_vs_reference_device_kernels.py tests/test_gpu_points_composite_interp.py -x -q -p no:cacheprovider ) > $O/tests.txt 2>&1; echo "tests rc=$?"
tail -6 $O/tests.txt
( time timeout 900 python bench.py ) > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"
tail -3 $O/bench.err
python - <<'PY'
import json
b=[l for l in open('gpurun_out/r06c8/bench.json') if l.startswith('{')]
b=json.loads(b[0])
print(round(b['value'],1),'Mpix/s',round(b['ms_per_step'],4),'ms; without prewarm', b.get('without_prewarm'))
print('vs_baseline', b.get('vs_baseline'), {k:v for k,v in (b.get('vs_reference_device') or {}).items() if k in ('value','forward_ms','backward_ms','speedup','error','reason')})
print('roofline', {k:b['roofline'][k] for k in ('bound','frac','valu','lds')})
oc=b.get('other_configs',{})
print('config5', oc.get('config5_jobs512_1gpu'))
print('config4 cpu', (oc.get('config4_points_1m_512_k10_fwd_bwd') or {}).get('cpu_baseline'))
print('cpu_baseline', {k:v for k,v in (b.get('cpu_baseline') or {}).items() if k!='python_reference'})
print('dropin', b['config'].get('dropin_ms_per_step'))
PY

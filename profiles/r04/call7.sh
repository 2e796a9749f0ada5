#!/bin/bash
# Round 4, GPU call 7: the injected-RCCL-failure fallback; soft-Phong pipeline product vs the bucket-probing variant of the
# shading tables; SQ / LDS counters of soft_phong_bwd (what binds the largest kernel of the end-to-end step).
set -u
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
O=gpurun_out/r04c7
mkdir -p $O
L=$PWD/pytorch3d_amd
stamp() { echo "== $1 $(date +%T)" | tee -a $O/steps.txt; }
stamp fallback
timeout 300 python -m pytest tests/test_gpu_bench_contract.py -q -x -k "rccl_failure or two_ranks" 2>&1 | tail -3
stamp pipeline
timeout 120 python profiles/bench_pipeline.py > $O/pipeline_product.json 2> $O/pipeline_product.err; cut -c1-900 $O/pipeline_product.json
P3D_LIB_PATH=$L/libp3d_buckets.so timeout 120 python profiles/bench_pipeline.py > $O/pipeline_buckets.json 2> $O/pipeline_buckets.err; cut -c1-900 $O/pipeline_buckets.json
stamp pmc
timeout 200 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_LDS \
  --kernel-trace --output-format csv -d $O/prof/pmc_sq -- python profiles/bench_pipeline.py > $O/pmc_sq.log 2>&1
timeout 200 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_VMEM \
  --kernel-trace --output-format csv -d $O/prof/pmc_lds -- python profiles/bench_pipeline.py > $O/pmc_lds.log 2>&1
find $O/prof -type f ! -name "*.csv" -delete
python - <<PY
import csv,glob,collections
for d in ("pmc_sq","pmc_lds"):
    f=glob.glob("$O/prof/%s/*/*_counter_collection.csv"%d)
    if not f: print(d,"no csv"); continue
    acc=collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(f[0])):
        n=r["Kernel_Name"]
        for k in ("soft_phong_bwd","soft_phong_fwd","phong_bwd","softmax_blend_bwd","mesh_backward","interp"):
            if k in n:
                acc[k][r["Counter_Name"]].append(float(r["Counter_Value"])); break
    for k,cs in acc.items():
        print(k, {c: "%.3g"%(sum(v)/len(v)) for c,v in sorted(cs.items())})
PY
stamp end

#!/bin/bash
# Round 4, GPU call 1: the literal SURVEY 8(d) config 3 workload (tori unscaled) as the baseline of the round.
#   1. product vs the four prepared variants of round 3 (profiles/next/*.patch) in one process, both workloads
#   2. the new sentinel test + the full-batch parity test on both workloads
#   3. bench.py short line (kernel breakdown on the literal workload)   4. K sweeps product vs rows4   5. rocprof SQ pass
set -u
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
O=gpurun_out/r04c1
mkdir -p $O
L=$PWD/pytorch3d_amd
stamp() { echo "== $1 $(date +%T)" | tee -a $O/steps.txt; }
stamp exp_literal
timeout 200 python profiles/exp_measure.py --torus-div 1.0 pixmask=$L/libp3d_pixmask.so pixpairs=$L/libp3d_pixpairs.so rows4=$L/libp3d_rows4.so \
  > $O/exp_literal.jsonl 2> $O/exp_literal.txt; tail -n 6 $O/exp_literal.txt
stamp exp_light
timeout 200 python profiles/exp_measure.py --torus-div 1.5 pixmask=$L/libp3d_pixmask.so pixpairs=$L/libp3d_pixpairs.so \
  > $O/exp_light.jsonl 2> $O/exp_light.txt; tail -n 5 $O/exp_light.txt
stamp tests
timeout 400 python -m pytest tests/test_gpu_meshes.py tests/test_gpu_bench_launch_parity.py tests/test_gpu_cover.py -x -q -s --durations=8 > $O/tests.txt 2>&1
echo "rc=$?" >> $O/tests.txt; grep -E "^\[bench|passed|failed|rc=|Error|assert" $O/tests.txt | cut -c1-400 | tail -12
stamp bench
timeout 200 python bench.py --steps 50 --no-cpu-baseline --no-dropin > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"
python - <<PY
import json
b=json.load(open("$O/bench.json"))
print(round(b["value"],1), "Mpix/s", round(b["ms_per_step"],4), "ms", b["kernels_ms"])
print("roofline", {k:(round(v["avg_ms"],4), round(v["algorithmic_gbps"]/8000,3)) for k,v in b["roofline"]["per_kernel"].items()})
print("cfg", {k:b["config"][k] for k in ("pixel_slot_fill","covered_pixel_fraction","total_faces_per_rank")})
print("light", b.get("workload_torus_div_1.5"))
print({k:(x.get("wall_ms"),x.get("kernels_ms")) for k,x in b["other_configs"].items()})
PY
stamp ksweep
timeout 120 python profiles/k_sweep.py 12 16 32 40 64 100 > $O/k_product.txt 2>&1; cat $O/k_product.txt
P3D_LIB_PATH=$L/libp3d_rows4.so timeout 120 python profiles/k_sweep.py 12 16 32 40 64 100 > $O/k_rows4.txt 2>&1; cat $O/k_rows4.txt
stamp rocprof
BENCH="python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-other-configs --no-dropin"
timeout 150 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_LDS \
  --kernel-trace --output-format csv -d $O/prof/pmc_sq -- $BENCH > $O/pmc_sq.log 2>&1
timeout 150 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_VMEM \
  --kernel-trace --output-format csv -d $O/prof/pmc_lds -- $BENCH > $O/pmc_lds.log 2>&1
find $O/prof -type f ! -name "*.csv" -delete
python profiles/summarize.py $O/prof $O/c1 1.0 > /dev/null 2>&1
grep -E "^###|VALU wave|SQ_INSTS_VALU|SQ_LDS_BANK|SQ_LDS_IDX|SQ_WAVES " $O/c1_rocprof.md | head -40
stamp end

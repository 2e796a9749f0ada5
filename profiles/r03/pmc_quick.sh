#!/bin/bash
# one SQ counter pass + kernel trace of a short bench run (round-3 working profile; the full set is profiles/run_rocprof.sh)
set -u
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
OUT=gpurun_out/r03/pmc
rm -rf $OUT; mkdir -p $OUT
BENCH="python bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-other-configs --no-dropin"
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_LDS \
  --kernel-trace --output-format csv -d $OUT/pmc_sq -- $BENCH > $OUT/pmc_sq.log 2>&1
rocprofv3 --pmc SQ_INSTS_SALU SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE GRBM_COUNT \
  --kernel-trace --output-format csv -d $OUT/pmc_2 -- $BENCH > $OUT/pmc_2.log 2>&1
python - <<'PY'
import csv, glob, collections
for d in ("pmc_sq", "pmc_2"):
    f = glob.glob(f"gpurun_out/r03/pmc/{d}/*/*_counter_collection.csv")
    if not f: print(d, "no csv"); continue
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(f[0])):
        k = r["Kernel_Name"]
        name = "mesh_fine" if "mesh_raster_kernel" in k else ("mesh_backward" if "mesh_backward" in k else None)
        if name: acc[name][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for name, cs in acc.items():
        print(name, {c: f"{sum(v)/len(v):.4g}" for c, v in sorted(cs.items())})
f = glob.glob("gpurun_out/r03/pmc/pmc_sq/*/*_kernel_trace.csv")
if f:
    dur = collections.defaultdict(list)
    for r in csv.DictReader(open(f[0])):
        k = r["Kernel_Name"]
        name = "mesh_fine" if "mesh_raster_kernel" in k else ("mesh_backward" if "mesh_backward" in k else None)
        if name: dur[name].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
    for n, v in dur.items(): print(n, "avg us", sum(v) / len(v), "n", len(v))
PY
find $OUT -type f ! -name "*.csv" ! -name "*.log" -delete

#!/bin/bash
# Round 3, GPU call 9: SQ counters of mesh_fine for the three builds of call 8 (base = previous commit, filter = depth
# bound, product = bound + interleaved queue moves): did the instruction count fall, and where did the cycles go?
set -u
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
O=gpurun_out/r03c9
mkdir -p $O
L=$PWD/pytorch3d_amd
BENCH="python bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-other-configs --no-dropin"
for v in base filter amd; do
  P3D_LIB_PATH=$L/libp3d_$v.so rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_SALU \
    --kernel-trace --output-format csv -d $O/$v -- $BENCH > $O/$v.log 2>&1
done
for v in base filter; do
  P3D_LIB_PATH=$L/libp3d_$v.so rocprofv3 --pmc SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_ANY SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_INSTS_SMEM SQ_ACTIVE_INST_MISC GRBM_GUI_ACTIVE \
    --kernel-trace --output-format csv -d $O/${v}_2 -- $BENCH > $O/${v}_2.log 2>&1
done
python - <<'PY'
import csv, glob, collections
for d in ("base", "filter", "amd", "base_2", "filter_2"):
    f = glob.glob(f"gpurun_out/r03c9/{d}/*/*_counter_collection.csv")
    if not f: print(d, "no csv"); continue
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(f[0])):
        if "mesh_raster_kernel" in r["Kernel_Name"]: acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
    print(d, {c: f"{sum(v)/len(v):.4g}" for c, v in sorted(acc.items())})
    f = glob.glob(f"gpurun_out/r03c9/{d}/*/*_kernel_trace.csv")
    dur = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3 for r in csv.DictReader(open(f[0])) if "mesh_raster_kernel" in r["Kernel_Name"]]
    print("   mesh_fine avg us", sum(dur) / len(dur), "n", len(dur))
PY
find $O -type f ! -name "*.csv" ! -name "*.log" -delete

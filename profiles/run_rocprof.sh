#!/bin/bash
# Collect the rocprofv3 evidence for bench.py on the GPU box (run from the repo root via gpurun).
#   1. kernel trace + stats  (per-kernel average duration; must agree with bench.py's HIP-event timing)
#   2..4. PMC passes, each in its own run with --kernel-trace only (SQ counters, FETCH_SIZE, WRITE_SIZE)
set -u
export TMPDIR=/tmp
OUT=${1:-gpurun_out/prof}
mkdir -p "$OUT"
# $2 (optional): another command to profile instead of the bench step, e.g. "python profiles/dropin_points_timing.py --mode patched"
BENCH=${2:-"python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-other-configs --no-dropin --no-reference-device"}
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats" -- $BENCH > "$OUT/stats.log" 2>&1
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_LDS \
  --kernel-trace --output-format csv -d "$OUT/pmc_sq" -- $BENCH > "$OUT/pmc_sq.log" 2>&1
rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_VMEM \
  --kernel-trace --output-format csv -d "$OUT/pmc_lds" -- $BENCH > "$OUT/pmc_lds.log" 2>&1
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d "$OUT/pmc_fetch" -- $BENCH > "$OUT/pmc_fetch.log" 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d "$OUT/pmc_write" -- $BENCH > "$OUT/pmc_write.log" 2>&1
find "$OUT" -name "*.csv" | head -50
# keep the merge small: drop anything that is not a csv/log
find "$OUT" -type f ! -name "*.csv" ! -name "*.log" -delete
du -sh "$OUT"

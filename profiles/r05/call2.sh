#!/bin/bash
# round 5, call 2: config 4 through the unmodified PointsRenderer (test + rocprofv3 kernel trace + SQ / FETCH / WRITE counter passes),
# and the driver's command `python bench.py` on the round's first build (baseline for the round's kernel work).
set -u
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
O=gpurun_out/r05c2
mkdir -p $O
stamp() { echo "== $1 $(date +%T)" | tee -a $O/steps.txt; }
stamp test
timeout 300 python -m pytest tests/test_gpu_points_renderer_dropin.py -x -q -s -p no:cacheprovider > $O/test_points_dropin.txt 2>&1; echo "test rc=$?"
tail -3 $O/test_points_dropin.txt
stamp bench
timeout 600 python bench.py > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"
cut -c1-300 $O/bench.json
stamp rocprof_c4
C4="python profiles/dropin_points_timing.py --steps 20"
rocprofv3 --kernel-trace --stats --output-format csv -d $O/c4/stats -- $C4 > $O/c4_stats.log 2>&1
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_LDS \
  --kernel-trace --output-format csv -d $O/c4/pmc_sq -- $C4 > $O/c4_pmc_sq.log 2>&1
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/c4/pmc_fetch -- $C4 > $O/c4_pmc_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/c4/pmc_write -- $C4 > $O/c4_pmc_write.log 2>&1
find $O/c4 -type f ! -name "*.csv" -delete
du -sh $O
stamp end

#!/bin/bash
# round 6, call 36: rocprofv3 kernel trace + PMC passes of the patched (fused) PointsRenderer step (config 4 as written)
set -u
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
O=gpurun_out/r06c36
mkdir -p $O
timeout 900 bash profiles/run_rocprof.sh $O/prof "python profiles/dropin_points_timing.py --mode patched --steps 20 --prewarm-s 0" > $O/rocprof.log 2>&1
tail -n 3 $O/rocprof.log

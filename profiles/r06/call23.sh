#!/bin/bash
# round 6, call 23: what the patched (fused) PointsRenderer step launches (kernel trace of 20 steps)
set -u
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
O=gpurun_out/r06c23
mkdir -p $O
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- python profiles/dropin_points_timing.py --mode patched --steps 20 > $O/stats.log 2>&1
tail -n 2 $O/stats.log | cut -c 1-600
find $O -type f ! -name "*kernel_stats.csv" ! -name "*.log" -delete
python - <<'PY'
import csv,glob
for f in glob.glob('gpurun_out/r06c23/stats/**/*kernel_stats.csv', recursive=True):
    rows=list(csv.DictReader(open(f)))
    for r in rows[:40]:
        print(r['Name'][:90].ljust(90), r['Calls'], r['AverageNs'], r['Percentage'])
PY

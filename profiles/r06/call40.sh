#!/bin/bash
# round 6, call 40: address-translation counters of mesh_fine per allocation (slow vs fast class)
set -u
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
O=gpurun_out/r06c40
mkdir -p $O
true
timeout 600 rocprofv3 --pmc TCP_UTCL1_REQUEST_sum TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_TRANSLATION_HIT_sum --kernel-trace --output-format csv -d $O/pmc -- python profiles/placement_probe.py --only-allocations --iters 5 > $O/run.txt 2> $O/run.err
tail -n 7 $O/run.txt; tail -n 3 $O/run.err | cut -c 1-300
find $O/pmc -name "*counter_collection.csv" | head

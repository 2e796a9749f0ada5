#!/bin/bash
# round 6, call 25: the reduced reproducer of round 4's lost queue entries (spilled lane table + readlane of an inactive lane)
set -u
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
mkdir -p gpurun_out/r06c25
hipcc --offload-arch=gfx950 -O3 -Wno-unused-result -Wno-unused-value profiles/microbench/spill_readlane.hip -o /tmp/spill_readlane 2> gpurun_out/r06c25/build.err
timeout 60 /tmp/spill_readlane | tee gpurun_out/r06c25/spill_readlane_mi355x.txt

#!/bin/bash
# round 6, call 3: A/B of the insertion network that runs only its live steps (variant "ins") against the product
set -u
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
bash profiles/ab_call.sh ins
mkdir -p gpurun_out/r06c3; cp -r gpurun_out/ab/* gpurun_out/r06c3/

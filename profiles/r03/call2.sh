#!/bin/bash
set -u
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
O=gpurun_out/r03
mkdir -p $O
rocminfo | grep -E "Marketing Name|Compute Unit|Max Clock" | tail -4 > $O/device.txt 2>&1
rocm-smi --showclocks 2>/dev/null | head -20 >> $O/device.txt
timeout 300 ./profiles/microbench/valu_issue.bin > $O/valu_issue_mi355x.txt 2>&1; cat $O/valu_issue_mi355x.txt | grep -E 'cndmask|exec|^#' | cut -c1-230
timeout 600 python -m pytest tests/test_gpu_bench_launch_parity.py -x -q -s 2>&1 | grep -E 'bench launch|passed|failed'

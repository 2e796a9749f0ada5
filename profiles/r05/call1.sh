#!/bin/bash
# round 5, call 1: the whole -m gpu suite on a fresh box with every duration (where do the seconds go?)
mkdir -p gpurun_out/r05c1
nproc > gpurun_out/r05c1/box.txt; cat /sys/fs/cgroup/cpu.max >> gpurun_out/r05c1/box.txt 2>&1; python -c "import os;print(len(os.sched_getaffinity(0)))" >> gpurun_out/r05c1/box.txt
( time python -m pytest tests -x -q -m gpu --durations=0 -p no:cacheprovider ) > gpurun_out/r05c1/tests_full.txt 2>&1
echo "rc=$?" >> gpurun_out/r05c1/tests_full.txt
tail -5 gpurun_out/r05c1/tests_full.txt

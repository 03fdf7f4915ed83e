#!/bin/bash
# usage: tools/gpu_prof_cmd.sh <tag> <divisor> <command...>: rocprofv3 kernel statistics of a command -> gpurun_out/kernel_stats_<tag>.txt (top 45 printed)
TAG=$1; shift; export TMPDIR=/tmp; mkdir -p gpurun_out
(cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/prof_$TAG -- "$@" > $GRAFT_REPO_ROOT/gpurun_out/rocprof_$TAG.log 2>&1)
python tools/rocprof_summary.py gpurun_out/prof_$TAG gpurun_out/kernel_stats_$TAG.txt "$TAG" > /dev/null && rm -rf gpurun_out/prof_$TAG
head -50 gpurun_out/kernel_stats_$TAG.txt | cut -c1-150

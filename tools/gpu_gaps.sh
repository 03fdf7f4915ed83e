#!/bin/bash
# usage: tools/gpu_gaps.sh <tag> : GPU idle time per training step (two-stream timed configuration) from a rocprofv3 kernel trace of bench.py
TAG=${1:-gaps}; export TMPDIR=/tmp; mkdir -p gpurun_out
(cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/prof_$TAG -- python $GRAFT_REPO_ROOT/bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-profile --no-infer --no-fp32 --no-free-run > $GRAFT_REPO_ROOT/gpurun_out/rocprof_$TAG.log 2>&1)
python tools/gap_analysis.py gpurun_out/prof_$TAG gpurun_out/step_gaps_$TAG.json | tail -12
rm -rf gpurun_out/prof_$TAG

#!/bin/bash
# usage: tools/gpu_step_ab.sh <tag> "<ENV=..>" "<ENV=..>" ... : the bench step under each environment setting on ONE box, interleaved twice:
# img/s of the un-instrumented loop, then rocprofv3 kernel statistics per setting (tools/kcat.py table)
TAG=$1; shift; mkdir -p gpurun_out; export TMPDIR=/tmp
for rep in 1 2; do
  for e in "$@"; do
    echo -n "[$rep] $e : "
    env $e timeout 600 python bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-profile --no-infer --no-fp32 2>/dev/null | tail -1 | cut -c1-120
  done
done
i=0
for e in "$@"; do
  (cd /tmp && env $e timeout 600 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/prof_${TAG}_$i -- python $GRAFT_REPO_ROOT/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-profile --no-infer --no-fp32 > $GRAFT_REPO_ROOT/gpurun_out/rocprof_${TAG}_$i.log 2>&1)
  python tools/rocprof_summary.py gpurun_out/prof_${TAG}_$i gpurun_out/kernel_stats_${TAG}_$i.txt "$TAG $e" > /dev/null && rm -rf gpurun_out/prof_${TAG}_$i
  i=$((i+1))
done
python tools/kcat.py gpurun_out/kernel_stats_${TAG}_*.txt

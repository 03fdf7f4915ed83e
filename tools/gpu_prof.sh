#!/bin/bash
# usage: tools/gpu_prof.sh <tag> : tests (quick), conv microbench, bench line, rocprof kernel stats -> gpurun_out/
TAG=${1:-x}
mkdir -p gpurun_out; export TMPDIR=/tmp
python -m pytest tests -m gpu -q --tb=line -p no:cacheprovider -x 2>&1 | tail -3
python tools/bench_conv.py --dtype bf16 --iters 3 > gpurun_out/bench_conv_$TAG.log 2>&1; tail -1 gpurun_out/bench_conv_$TAG.log
python bench.py --steps 4 --warmup 2 --no-cpu-baseline > gpurun_out/bench_$TAG.log 2>&1; tail -1 gpurun_out/bench_$TAG.log | cut -c1-250
cd /tmp && rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/prof_$TAG -- python $GRAFT_REPO_ROOT/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-profile --no-infer > $GRAFT_REPO_ROOT/gpurun_out/rocprof_$TAG.log 2>&1
cd $GRAFT_REPO_ROOT && python tools/rocprof_summary.py gpurun_out/prof_$TAG gpurun_out/kernel_stats_$TAG.txt "$TAG: bench.py --steps 2 --warmup 1 (3 train steps, 512x512 b16 bf16)" && rm -rf gpurun_out/prof_$TAG && head -45 gpurun_out/kernel_stats_$TAG.txt | cut -c1-150

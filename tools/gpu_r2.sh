#!/bin/bash
# usage: tools/gpu_r2.sh <tag> [full] : GPU parity tests (full: whole suite; else stop at first failure), bench line, rocprof kernel stats
TAG=${1:-x}; MODE=${2:-quick}
mkdir -p gpurun_out; export TMPDIR=/tmp
if [ "$MODE" = full ]; then
  timeout 2400 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider > gpurun_out/pytest_$TAG.log 2>&1; echo "pytest rc=$?" >> gpurun_out/pytest_$TAG.log
else
  timeout 2400 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider -x > gpurun_out/pytest_$TAG.log 2>&1; echo "pytest rc=$?" >> gpurun_out/pytest_$TAG.log
fi
tail -25 gpurun_out/pytest_$TAG.log | cut -c1-220
timeout 600 python bench.py --steps 6 --warmup 2 --no-cpu-baseline > gpurun_out/bench_$TAG.log 2>&1; tail -1 gpurun_out/bench_$TAG.log | cut -c1-600
cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/prof_$TAG -- python $GRAFT_REPO_ROOT/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-profile --no-infer > $GRAFT_REPO_ROOT/gpurun_out/rocprof_$TAG.log 2>&1
cd $GRAFT_REPO_ROOT && python tools/rocprof_summary.py gpurun_out/prof_$TAG gpurun_out/kernel_stats_$TAG.txt "$TAG: bench.py --steps 2 --warmup 1 (3 train steps, 512x512 b16 bf16)" && rm -rf gpurun_out/prof_$TAG && head -60 gpurun_out/kernel_stats_$TAG.txt | cut -c1-150
cat gpurun_out/bf16_deviation.json 2>/dev/null

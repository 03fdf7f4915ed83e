#!/bin/bash
# usage: tools/gpu_quick.sh <tag> [pytest -k expr] : selected GPU tests, then a short bench line -> gpurun_out/
TAG=${1:-q}; K=${2:-}
mkdir -p gpurun_out; export TMPDIR=/tmp
if [ -n "$K" ]; then
  timeout 1500 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider -k "$K" > gpurun_out/pytest_$TAG.log 2>&1; echo "pytest rc=$?" >> gpurun_out/pytest_$TAG.log
else
  timeout 2400 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider > gpurun_out/pytest_$TAG.log 2>&1; echo "pytest rc=$?" >> gpurun_out/pytest_$TAG.log
fi
tail -30 gpurun_out/pytest_$TAG.log | cut -c1-400
timeout 600 python bench.py --steps 8 --warmup 2 --no-cpu-baseline --no-fp32 > gpurun_out/bench_$TAG.log 2>&1; tail -1 gpurun_out/bench_$TAG.log | cut -c1-700

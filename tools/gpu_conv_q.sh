#!/bin/bash
# usage: tools/gpu_conv_q.sh <tag> <filter> [pytest -k expr]: selected GPU tests, per-layer conv table (batch 16 and 32), D-path timing, short bench line
TAG=$1; FILT=$2; K=${3:-}
mkdir -p gpurun_out; export TMPDIR=/tmp
if [ -n "$K" ]; then
  timeout 1500 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider -x -k "$K" > gpurun_out/pytest_$TAG.log 2>&1; echo "pytest rc=$?" >> gpurun_out/pytest_$TAG.log
  tail -4 gpurun_out/pytest_$TAG.log | cut -c1-300
fi
timeout 600 python tools/bench_conv.py --batch 16 --iters 10 --filter "$FILT" 2>/dev/null | grep -v "^layer" | cut -c1-110
timeout 600 python tools/bench_conv.py --batch 32 --iters 10 --filter "$FILT" 2>/dev/null | grep -v "^layer" | cut -c1-110
timeout 300 python tools/bench_d.py 2>/dev/null | tail -2
timeout 600 python bench.py --steps 8 --warmup 2 --no-cpu-baseline --no-profile --no-infer --no-fp32 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('step ms', d['ms_per_step'], 'no_readback', d['no_readback']['ms_per_step'])"

#!/bin/bash
# usage: tools/gpu_ab_tune.sh <tag> <filter> <tuneA> <tuneB> [pytest -k expr]: per-layer conv table with two tuning settings + step time with each, on ONE box
TAG=$1; FILT=$2; TA=$3; TB=$4; K=${5:-}
mkdir -p gpurun_out; export TMPDIR=/tmp
if [ -n "$K" ]; then
  timeout 1200 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider -x -k "$K" > gpurun_out/pytest_$TAG.log 2>&1; echo "pytest rc=$?" >> gpurun_out/pytest_$TAG.log
  tail -4 gpurun_out/pytest_$TAG.log | cut -c1-300
fi
for T in "$TA" "$TB"; do
  echo "== tune $T"
  timeout 600 python tools/bench_conv.py --batch 16 --iters 10 --filter "$FILT" --tune "$T" 2>/dev/null | grep -v "^layer" | cut -c1-110
  timeout 600 python tools/bench_conv.py --batch 32 --iters 10 --filter "$FILT" --tune "$T" 2>/dev/null | grep -v "^layer" | cut -c1-110
  timeout 600 python bench.py --tune "$T" --steps 8 --warmup 2 --no-cpu-baseline --no-profile --no-infer --no-fp32 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('step ms', d['ms_per_step'], 'no_readback', d['no_readback']['ms_per_step'], 'host', d['host'])"
done

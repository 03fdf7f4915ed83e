#!/bin/bash
# usage: tools/gpu_ab_flag.sh <tag> "<bench flags A>" "<bench flags B>" [pytest -k expr]: the same bench line with two flag sets on ONE box (interleaved, twice each)
TAG=$1; FA=$2; FB=$3; K=${4:-}
mkdir -p gpurun_out; export TMPDIR=/tmp
if [ -n "$K" ]; then
  timeout 1500 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider -x -k "$K" > gpurun_out/pytest_$TAG.log 2>&1; echo "pytest rc=$?" >> gpurun_out/pytest_$TAG.log
  tail -4 gpurun_out/pytest_$TAG.log | cut -c1-300
fi
for rep in 1 2; do
  for F in "$FA" "$FB"; do
    echo "== [$rep] flags: $F"
    timeout 600 python bench.py $F --steps 10 --warmup 3 --no-cpu-baseline --no-profile --no-infer --no-fp32 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('step ms', d['ms_per_step'], 'no_readback', d['no_readback']['ms_per_step'], 'img/s', d['value'])"
  done
done

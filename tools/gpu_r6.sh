#!/bin/bash
# usage: tools/gpu_r6.sh <tag> [pytest -k expr]: selected GPU tests, the bench line with the fp32 / fp16 / fp16-precise legs, one-stream kernel statistics
TAG=${1:-r6}; K=${2:-}
mkdir -p gpurun_out; export TMPDIR=/tmp
if [ -n "$K" ]; then
  timeout 2400 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider -x -k "$K" > gpurun_out/pytest_$TAG.log 2>&1; echo "pytest rc=$?" >> gpurun_out/pytest_$TAG.log
  tail -25 gpurun_out/pytest_$TAG.log | cut -c1-400
fi
timeout 900 python bench.py --steps 10 --warmup 3 --no-cpu-baseline > gpurun_out/bench_$TAG.log 2>&1
tail -1 gpurun_out/bench_$TAG.log | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print('bf16', d['value'], d['ms_per_step'], 'free', d['no_readback']['ms_per_step'], 'host', d['host']['issue_ms_per_step'], d['host']['abi_calls_per_step'])
for k in ('fp32','fp16','fp16_precise'):
    if k in d: print(k, d[k]['value'], d[k]['ms_per_step'], d[k].get('infer_ms_per_img'))
print('infer', d.get('infer_ms_per_img'), d['infer']['batch8']['ms_per_img'])
r=d['roofline']; print('roof', r['kernel'], r['frac'], 'hbm', r['hbm_kernel']['kernel'], r['hbm_kernel']['frac'])
for t in r['top5']: print('  ', t)
"
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/prof_$TAG -- python $GRAFT_REPO_ROOT/bench.py --one-stream --steps 2 --warmup 1 --no-cpu-baseline --no-profile --no-infer --no-fp32 --no-free-run > $GRAFT_REPO_ROOT/gpurun_out/rocprof_$TAG.log 2>&1)
python tools/rocprof_summary.py gpurun_out/prof_$TAG gpurun_out/kernel_stats_$TAG.txt "$TAG: bench.py --one-stream --steps 2 --warmup 1 (3 train steps on ONE stream, 512x512 b16 bf16)" > /dev/null && rm -rf gpurun_out/prof_$TAG
head -30 gpurun_out/kernel_stats_$TAG.txt | cut -c1-200

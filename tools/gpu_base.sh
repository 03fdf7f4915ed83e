#!/bin/bash
# usage: tools/gpu_base.sh <tag> : full GPU tests, smoke, short bench line, one-stream kernel statistics, per-layer conv table, D-path timing
TAG=${1:-base}; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 2400 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider -x > gpurun_out/pytest_$TAG.log 2>&1; echo "pytest rc=$?" >> gpurun_out/pytest_$TAG.log
tail -5 gpurun_out/pytest_$TAG.log | cut -c1-300
timeout 600 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
timeout 900 python bench.py --no-cpu-baseline > gpurun_out/bench_${TAG}.log 2>&1; tail -1 gpurun_out/bench_${TAG}.log | cut -c1-400
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/prof_$TAG -- python $GRAFT_REPO_ROOT/bench.py --one-stream --steps 2 --warmup 1 --no-cpu-baseline --no-profile --no-infer --no-fp32 > $GRAFT_REPO_ROOT/gpurun_out/rocprof_$TAG.log 2>&1)
python tools/rocprof_summary.py gpurun_out/prof_$TAG gpurun_out/kernel_stats_$TAG.txt "$TAG: bench.py --one-stream --steps 2 --warmup 1 (5 train steps on ONE stream, 512x512 b16 bf16)" > /dev/null && rm -rf gpurun_out/prof_$TAG
timeout 600 python tools/bench_conv.py --batch 16 --iters 10 > gpurun_out/bench_conv_$TAG.log 2>&1; tail -1 gpurun_out/bench_conv_$TAG.log
timeout 300 python tools/bench_d.py > gpurun_out/bench_d_$TAG.log 2>&1; tail -2 gpurun_out/bench_d_$TAG.log

#!/bin/bash
# usage: tools/gpu_final.sh <tag> : the round's record at HEAD -- full GPU test suite, smoke(), the default bench line (with the CPU baseline), rocprof kernel
# statistics of the same command, and the extra bench lines (fp32, 1024^2 batch 8, per-line passes)
TAG=${1:-final}; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 2400 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider > gpurun_out/pytest_$TAG.log 2>&1; echo "pytest rc=$?" >> gpurun_out/pytest_$TAG.log
tail -4 gpurun_out/pytest_$TAG.log | cut -c1-200
timeout 600 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
timeout 900 python bench.py > gpurun_out/bench_${TAG}_default.log 2>&1; tail -1 gpurun_out/bench_${TAG}_default.log | cut -c1-300
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/prof_$TAG -- python $GRAFT_REPO_ROOT/bench.py --one-stream --steps 2 --warmup 1 --no-cpu-baseline --no-profile --no-infer --no-fp32 --no-free-run > $GRAFT_REPO_ROOT/gpurun_out/rocprof_$TAG.log 2>&1)
python tools/rocprof_summary.py gpurun_out/prof_$TAG gpurun_out/kernel_stats_$TAG.txt "$TAG: bench.py --one-stream --steps 2 --warmup 1 (3 train steps on ONE stream, 512x512 b16 bf16)" > /dev/null && rm -rf gpurun_out/prof_$TAG
timeout 900 python bench.py --dtype f32 --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/bench_${TAG}_f32.log 2>&1; tail -1 gpurun_out/bench_${TAG}_f32.log | cut -c1-200
timeout 900 python bench.py --size 1024 --batch 8 --steps 4 --warmup 2 --no-cpu-baseline > gpurun_out/bench_${TAG}_1024_b8.log 2>&1; tail -1 gpurun_out/bench_${TAG}_1024_b8.log | cut -c1-200
timeout 900 python bench.py --per-line --steps 6 --warmup 2 --no-cpu-baseline --no-infer > gpurun_out/bench_${TAG}_per_line.log 2>&1; tail -1 gpurun_out/bench_${TAG}_per_line.log | cut -c1-200
timeout 600 python tools/bench_conv.py --batch 16 --iters 10 > gpurun_out/bench_conv_$TAG.log 2>&1; tail -1 gpurun_out/bench_conv_$TAG.log
# round 6: the precise fp16 mode's one-stream kernel statistics, the batch-1 inference timeline (bf16 and precise fp16)
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/prof_p_$TAG -- python $GRAFT_REPO_ROOT/bench.py --dtype f16 --precise --one-stream --steps 2 --warmup 1 --no-cpu-baseline --no-profile --no-infer --no-fp32 --no-free-run > $GRAFT_REPO_ROOT/gpurun_out/rocprof_p_$TAG.log 2>&1)
python tools/rocprof_summary.py gpurun_out/prof_p_$TAG gpurun_out/kernel_stats_f16_precise_$TAG.txt "$TAG: bench.py --dtype f16 --precise --one-stream --steps 2 --warmup 1 (3 train steps on ONE stream, fp16 storage, set_precise)" > /dev/null && rm -rf gpurun_out/prof_p_$TAG
for M in bf16 f16p; do
  (cd /tmp && timeout 300 rocprofv3 --kernel-trace -d $GRAFT_REPO_ROOT/gpurun_out/prof_i_$M -- python $GRAFT_REPO_ROOT/tools/infer_graph.py $M 20 > /dev/null 2>&1)
  python tools/infer_trace.py gpurun_out/prof_i_$M > gpurun_out/infer_trace_${M}_$TAG.txt; rm -rf gpurun_out/prof_i_$M; tail -1 gpurun_out/infer_trace_${M}_$TAG.txt
done

#!/bin/bash
# first GPU visit: parity tests, smoke, short benches, rocprof kernel stats
mkdir -p gpurun_out
export TMPDIR=/tmp
echo "=== rocminfo ===" > gpurun_out/info.log; (rocminfo | grep -E "Marketing|Compute Unit|gfx" | head -8; nproc; grep -m1 "model name" /proc/cpuinfo) >> gpurun_out/info.log 2>&1
timeout 1500 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider > gpurun_out/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> gpurun_out/pytest_gpu.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/smoke.log 2>&1; echo "smoke rc=$?" >> gpurun_out/smoke.log
timeout 900 python bench.py --steps 3 --warmup 1 --dtype bf16 --no-cpu-baseline --infer > gpurun_out/bench_bf16.log 2>&1; echo "rc=$?" >> gpurun_out/bench_bf16.log
timeout 900 python bench.py --steps 2 --warmup 1 --dtype f32 --no-cpu-baseline > gpurun_out/bench_f32.log 2>&1; echo "rc=$?" >> gpurun_out/bench_f32.log
cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/prof_bf16 -- python $GRAFT_REPO_ROOT/bench.py --steps 2 --warmup 1 --dtype bf16 --no-cpu-baseline --no-profile > $GRAFT_REPO_ROOT/gpurun_out/rocprof_bf16.log 2>&1
cd $GRAFT_REPO_ROOT
find gpurun_out/prof_bf16 -name "*kernel_stats*" | head -3
tail -3 gpurun_out/pytest_gpu.log; tail -3 gpurun_out/smoke.log; tail -2 gpurun_out/bench_bf16.log; tail -2 gpurun_out/bench_f32.log

#!/bin/bash
# the round's full record on ONE box: tools/gpu_final.sh (tests, smoke, bench lines, kernel statistics) + the step's PMC traffic
TAG=${1:-final}
bash tools/gpu_final.sh $TAG
bash tools/gpu_step_traffic.sh > gpurun_out/step_traffic_stdout.log 2>&1; head -12 gpurun_out/step_traffic.txt
bash tools/gpu_gaps.sh $TAG 2>&1 | tail -6
timeout 300 python tools/bench_d.py > gpurun_out/bench_d_$TAG.log 2>&1; tail -2 gpurun_out/bench_d_$TAG.log
cp gpurun_out/bf16_deviation.json gpurun_out/bf16_deviation_$TAG.json 2>/dev/null

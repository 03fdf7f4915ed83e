#!/bin/bash
# the round's full record on ONE box: tools/gpu_final.sh (tests, smoke, bench lines, kernel statistics) + the step's PMC traffic
TAG=${1:-final}
bash tools/gpu_final.sh $TAG
bash tools/gpu_step_traffic.sh > gpurun_out/step_traffic_stdout.log 2>&1; head -12 gpurun_out/step_traffic.txt

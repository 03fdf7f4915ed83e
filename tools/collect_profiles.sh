#!/bin/bash
# usage: tools/collect_profiles.sh <gpurun tag> <round prefix>: copy a tools/gpu_record.sh record from gpurun_out/ (scratch) into profiles/ (tracked)
TAG=${1:-r06_final}; R=${2:-r06}
P=profiles; G=gpurun_out
for n in default f32 1024_b8 per_line; do tail -1 $G/bench_${TAG}_$n.log > $P/${R}_final_bench_$n.json; done
cp $G/bench_conv_$TAG.log $P/${R}_final_bench_conv.log
cp $G/bench_d_$TAG.log $P/${R}_final_bench_d.log
cp $G/kernel_stats_$TAG.txt $P/${R}_final_kernel_stats.txt
cp $G/kernel_stats_f16_precise_$TAG.txt $P/${R}_final_kernel_stats_f16_precise.txt
cp $G/step_traffic.txt $P/${R}_final_step_traffic.txt
cp $G/step_gaps_$TAG.json $P/${R}_step_gaps.json
cp $G/bf16_deviation_$TAG.json $P/${R}_bf16_deviation.json
cp $G/f16_param_grads.json $P/${R}_f16_param_grads.json
cp $G/infer_trace_bf16_$TAG.txt $P/${R}_final_infer_trace_bf16.txt
cp $G/infer_trace_f16p_$TAG.txt $P/${R}_final_infer_trace_f16p.txt
tail -6 $G/pytest_$TAG.log > $P/${R}_final_pytest_tail.txt
python tools/update_pmc_json.py ${R}_final > /dev/null
ls -la $P | grep ${R}_ | wc -l

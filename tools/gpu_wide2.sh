#!/bin/bash
mkdir -p gpurun_out
for A in 0 1 2 3 4; do
  echo "== ABL=$A"
  UEGAN_WIDE_ABL=$A timeout 300 python tools/bench_conv.py --filter "VGG.conv9" --iters 10 2>&1 | grep "conv9"
done
bash tools/gpu_pmc2.sh "VGG.conv9" 16 conv_wide 2>&1 | tee gpurun_out/pmc_wide_conv9.txt

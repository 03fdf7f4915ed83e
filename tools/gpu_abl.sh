#!/bin/bash
# usage: tools/gpu_abl.sh : per-layer timing of the streaming / weight-gradient kernels under the timing ablations of the tools build (tools/build_tools.sh) (1 no staging after the
# first tile, 2 no K loop, 4 no stores) -- where the time of a thin layer goes
mkdir -p gpurun_out
for f in "G.dec4" "G.dec5" "G.enc1" "G.enc2" "G.dec3" "D.d3 " "D.d5 " "G.dec1"; do
  for abl in 0 1 2 3 4 6; do
    echo -n "ABL=$abl "; python tools/bench_conv.py --abl $abl --batch 32 --iters 6 --filter "$f" 2>&1 | grep -v "^layer\|TOTAL" | cut -c1-100
  done
done

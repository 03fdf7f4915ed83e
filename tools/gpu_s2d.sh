#!/bin/bash
# stride-2 data gradients: default / UEGAN_S2D=2 (8 x 16 tiles, two blocks per CU) / streaming class kernel off
for f in "G.enc2" "G.enc3" "G.enc4" "G.enc5" "D.d2 " "D.d3 " "D.d4 " "D.d5 "; do
  for v in "X=0" "UEGAN_S2D=2" "UEGAN_STREAM_NOCLS8=1"; do
    echo -n "$v "; env $v python tools/bench_conv.py --batch 32 --iters 6 --filter "$f" 2>&1 | grep -v "^layer\|TOTAL\|amdgpu.ids" | cut -c1-100
  done
done

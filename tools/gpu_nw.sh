#!/bin/bash
# usage: tools/gpu_nw.sh : thin layers with the streaming kernel's block size knob (UEGAN_STREAM_NW: 4 / default / 8)
for f in "G.dec4" "G.dec5" "G.enc1" "G.ga1" "VGG.conv0" "D.d1 " "D.d2 "; do
  for nw in 4 0 8; do
    echo -n "NW=$nw "; UEGAN_STREAM_NW=$nw python tools/bench_conv.py --batch 32 --iters 6 --filter "$f" 2>&1 | grep -v "^layer\|TOTAL\|amdgpu.ids" | cut -c1-100
  done
done

#!/bin/bash
for abl in 0 3 8 11; do
  echo -n "ABL=$abl "; UEGAN_ABL=$abl python tools/bench_conv.py --batch 32 --iters 6 --filter "G.dec5.1" 2>&1 | grep -v "^layer\|TOTAL\|amdgpu.ids" | cut -c1-130
done

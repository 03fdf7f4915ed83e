#!/bin/bash
# usage: tools/gpu_wg.sh : weight-gradient layers, default vs UEGAN_WGTR_NOSKIP=1 (stride-2 row skipping off)
timeout 600 python -m pytest tests/test_ops.py -m gpu -x -q -p no:cacheprovider -k "wgrad" 2>&1 | tail -2
for f in "G.dec5.1" "D.d2 " "D.d3 " "D.d4 " "D.d5 " "G.enc3" "D.d1_pred"; do
  echo -n "skip   "; python tools/bench_conv.py --batch 48 --iters 6 --filter "$f" 2>&1 | grep -v "^layer\|TOTAL\|amdgpu.ids" | cut -c1-130
  echo -n "noskip "; UEGAN_WGTR_NOSKIP=1 python tools/bench_conv.py --batch 48 --iters 6 --filter "$f" 2>&1 | grep -v "^layer\|TOTAL\|amdgpu.ids" | cut -c1-130
done

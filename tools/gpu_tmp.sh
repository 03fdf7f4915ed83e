#!/bin/bash
for v in "X=1" "UEGAN_S2HALF_2BUF=1" "X=2" "UEGAN_S2HALF_2BUF=1"; do
  echo -n "$v "; env $v python tools/bench_conv.py --batch 48 --iters 8 --filter "D.d2 " --kernels 2>&1 | grep "fwd " | cut -c1-160
done

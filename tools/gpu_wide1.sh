#!/bin/bash
# first contact of conv_wide.hip with the hardware: layout self-test, parity tests, per-layer A/B against conv_patch
mkdir -p gpurun_out
python -m pytest tests/test_ops.py -x -q -m gpu -k "wide or mfma_fragment" 2>&1 | tail -5
for W in 0 192; do
  echo "== UEGAN_WIDE=$W"
  UEGAN_WIDE=$W timeout 600 python tools/bench_conv.py --filter VGG.conv --iters 10 2>&1 | grep -v amdgpu.ids | tee gpurun_out/bench_conv_wide_$W.log | grep -E "conv(5|6|8|9|10|11|12) "
done

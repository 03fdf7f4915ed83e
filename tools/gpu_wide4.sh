#!/bin/bash
mkdir -p gpurun_out
python -m pytest tests/test_ops.py -x -q -m gpu -k "wide" 2>&1 | tail -2
for A in 0 4; do
  echo "== ABL=$A"
  UEGAN_WIDE_ABL=$A timeout 300 python tools/bench_conv.py --filter "VGG.conv9" --iters 10 2>&1 | grep "conv9"
done
UEGAN_WIDE=0 timeout 300 python tools/bench_conv.py --filter "VGG.conv9" --iters 10 2>&1 | grep "conv9"
for A in 0 4; do
UEGAN_WIDE_ABL=$A bash tools/gpu_pmc.sh "GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES" "conv_wide_kernel<3, 0" python $GRAFT_REPO_ROOT/tools/bench_conv.py --filter "VGG.conv9" --iters 3 2>&1 | tail -9
done

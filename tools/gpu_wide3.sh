#!/bin/bash
mkdir -p gpurun_out
for A in 0 4; do
  echo "== ABL=$A"
  UEGAN_WIDE_ABL=$A bash tools/gpu_pmc.sh "GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES" "conv_wide_kernel<3, 0" python $GRAFT_REPO_ROOT/tools/bench_conv.py --filter "VGG.conv9" --iters 3 2>&1 | tail -12
done
echo "== old kernel"
UEGAN_WIDE=0 bash tools/gpu_pmc.sh "GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES" "conv_patch_kernel<unsigned short, 256, 4, 2, 3, 0" python $GRAFT_REPO_ROOT/tools/bench_conv.py --filter "VGG.conv9" --iters 3 2>&1 | tail -12
rocm-smi --showclocks 2>&1 | head -30

#!/bin/bash
# usage: tools/gpu_pmc2.sh <layer filter> <batch> <kernel substring>: two PMC passes over tools/bench_conv.py for one layer
F="$1"; B="$2"; K="$3"
for CNT in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS" \
           "SQ_INSTS_MFMA SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_RD SQ_WAVES"; do
  bash tools/gpu_pmc.sh "$CNT" "$K" python $GRAFT_REPO_ROOT/tools/bench_conv.py --filter "$F" --batch $B --iters 2 2>&1 | tail -40
done

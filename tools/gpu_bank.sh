#!/bin/bash
# usage: tools/gpu_bank.sh "<layer filter>" ... : LDS bank-conflict counters of the kernels a layer of tools/bench_conv.py launches
for F in "$@"; do
  echo "=== $F"
  bash tools/gpu_pmc.sh "SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_WAVE_CYCLES SQ_INSTS_MFMA SQ_WAIT_ANY" "uegan" python $GRAFT_REPO_ROOT/tools/bench_conv.py --filter "$F" --batch 32 --iters 2 2>&1 | grep -v "^   duration\|amdgpu.ids" | cut -c1-110
done

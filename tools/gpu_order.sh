#!/bin/bash
python tools/bench_conv.py --batch 32 --iters 6 > gpurun_out/bc_order_default.log 2>&1
UEGAN_WGTR_ROWMAJOR=1 python tools/bench_conv.py --batch 32 --iters 6 > gpurun_out/bc_order_wrow.log 2>&1
UEGAN_STREAM_COLMAJOR=1 python tools/bench_conv.py --batch 32 --iters 6 > gpurun_out/bc_order_scol.log 2>&1
tail -1 gpurun_out/bc_order_*.log

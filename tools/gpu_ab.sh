#!/bin/bash
# usage: tools/gpu_ab.sh <tag> : A/B of uegan_amd/_ab/lib_base.so against the built library on ONE box (per-layer conv table + bench line)
TAG=${1:-ab}; mkdir -p gpurun_out; export TMPDIR=/tmp
cp uegan_amd/libuegan_hip.so /tmp/lib_new.so
for v in new base new base; do
  if [ $v = base ]; then cp uegan_amd/_ab/lib_base.so uegan_amd/libuegan_hip.so; else cp /tmp/lib_new.so uegan_amd/libuegan_hip.so; fi
  timeout 600 python tools/bench_conv.py --batch 16 --iters 10 > gpurun_out/convtab_${TAG}_$v.log 2>&1
  timeout 600 python bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-profile --no-infer 2>/dev/null | tail -1 | cut -c1-200 | tee -a gpurun_out/bench_${TAG}_$v.log
done
cp /tmp/lib_new.so uegan_amd/libuegan_hip.so
for v in new base; do echo "---- $v"; grep -E "VGG|G.dec|TOTAL|total" gpurun_out/convtab_${TAG}_$v.log | cut -c1-130; done

#!/bin/bash
# usage: tools/gpu_ab2.sh <tag> <libA> <libB> ... : rocprof kernel statistics of the bench step with each library variant on ONE box
TAG=$1; shift; mkdir -p gpurun_out; export TMPDIR=/tmp
cp uegan_amd/libuegan_hip.so /tmp/lib_head.so
for lib in head "$@"; do
  name=$(basename $lib .so)
  if [ $lib = head ]; then cp /tmp/lib_head.so uegan_amd/libuegan_hip.so; else cp $lib uegan_amd/libuegan_hip.so; fi
  (cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/prof_${TAG}_$name -- python $GRAFT_REPO_ROOT/bench.py --steps 2 --warmup 1 --one-stream --no-cpu-baseline --no-profile --no-infer --no-fp32 > $GRAFT_REPO_ROOT/gpurun_out/rocprof_${TAG}_$name.log 2>&1)
  python tools/rocprof_summary.py gpurun_out/prof_${TAG}_$name gpurun_out/kernel_stats_${TAG}_$name.txt "$TAG $name" > /dev/null && rm -rf gpurun_out/prof_${TAG}_$name
  timeout 600 python bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-profile --no-infer --no-fp32 2>/dev/null | tail -1 | cut -c1-160
done
cp /tmp/lib_head.so uegan_amd/libuegan_hip.so
python tools/kcat.py gpurun_out/kernel_stats_${TAG}_*.txt

#!/bin/bash
# usage: tools/gpu_sweep.sh "<ENVVAR=val ENVVAR2=val>" ... : rocprof kernel-time categories of the bench step under each environment setting
export TMPDIR=/tmp; mkdir -p gpurun_out
i=0
for setting in "$@"; do
  i=$((i+1))
  (cd /tmp && env $setting timeout 600 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/prof_sw$i -- python $GRAFT_REPO_ROOT/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-profile --no-infer > $GRAFT_REPO_ROOT/gpurun_out/rocprof_sw$i.log 2>&1)
  python tools/rocprof_summary.py gpurun_out/prof_sw$i gpurun_out/kernel_stats_sw$i.txt "$setting" > /dev/null && rm -rf gpurun_out/prof_sw$i
  echo "== sw$i: $setting"
done
python tools/kcat.py gpurun_out/kernel_stats_sw*.txt | grep -E "category|instnorm|percep|TOTAL"

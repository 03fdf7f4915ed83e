#!/bin/bash
# usage: tools/gpu_kstat.sh <command...> : rocprofv3 kernel-trace of a command, per-kernel stats to stdout
export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/kstat_tmp
rm -rf $OUT; mkdir -p $OUT
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats -d $OUT -- "$@" > $OUT/run.log 2>&1)
python $GRAFT_REPO_ROOT/tools/rocprof_summary.py $OUT $OUT/stats.txt "kstat" > /dev/null && head -${KSTAT_LINES:-24} $OUT/stats.txt | cut -c1-175

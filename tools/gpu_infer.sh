#!/bin/bash
# usage: tools/gpu_infer.sh <tag> : parity of the forward (pytest -k given as $2), then the batch-1 inference timeline (bf16) of the hipGraph replay
TAG=${1:-inf}; K=${2:-}
mkdir -p gpurun_out; export TMPDIR=/tmp
if [ -n "$K" ]; then
  timeout 1500 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider -x -k "$K" > gpurun_out/pytest_$TAG.log 2>&1; echo "pytest rc=$?" >> gpurun_out/pytest_$TAG.log
  tail -4 gpurun_out/pytest_$TAG.log | cut -c1-300
fi
for M in ${MODES:-bf16}; do
  (cd /tmp && timeout 300 rocprofv3 --kernel-trace -d $GRAFT_REPO_ROOT/gpurun_out/prof_i_$M -- python $GRAFT_REPO_ROOT/tools/infer_graph.py $M 20 > /dev/null 2>&1)
  python tools/infer_trace.py gpurun_out/prof_i_$M > gpurun_out/infer_trace_${M}_$TAG.txt; rm -rf gpurun_out/prof_i_$M; cut -c1-120 gpurun_out/infer_trace_${M}_$TAG.txt
done

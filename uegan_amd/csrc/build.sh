#!/bin/bash
# Build libuegan_hip.so for gfx950 (MI355X). hipcc cross-compiles without a GPU.
set -euo pipefail
HERE="$(cd "$(dirname "$0")" && pwd)"
OUT="${1:-$HERE/../libuegan_hip.so}"
HIPCC="${HIPCC:-/opt/rocm/bin/hipcc}"
SRCS=(conv.hip heads.hip elementwise.hip norm_loss.hip optim_sn.hip metrics.hip)
OBJS=()
mkdir -p "$HERE/_obj"
pids=()
for s in "${SRCS[@]}"; do
  o="$HERE/_obj/${s%.hip}.o"
  OBJS+=("$o")
  if [ ! -f "$o" ] || [ "$HERE/$s" -nt "$o" ] || [ "$HERE/common.h" -nt "$o" ] || [ "$HERE/conv_internal.h" -nt "$o" ] || [ "$HERE/wgrad_tr.h" -nt "$o" ] || [ "$HERE/conv_stream.h" -nt "$o" ] || [ "$HERE/../../include/uegan_hip.h" -nt "$o" ]; then
    "$HIPCC" --offload-arch=gfx950 -O3 -std=c++17 -fPIC -munsafe-fp-atomics -Wall -Wno-unused-function -c "$HERE/$s" -o "$o" &
    pids+=($!)
  fi
done
for p in "${pids[@]:-}"; do [ -n "$p" ] && wait "$p"; done
"$HIPCC" --offload-arch=gfx950 -shared -fPIC "${OBJS[@]}" -o "$OUT"
echo "built $OUT"

#!/bin/bash
# Build libuegan_hip.so for gfx950 (MI355X). hipcc cross-compiles without a GPU.  One object per translation unit, compiled in
# parallel; an object is rebuilt when its source or ANY header is newer.
set -euo pipefail
HERE="$(cd "$(dirname "$0")" && pwd)"
OUT="${1:-$HERE/../libuegan_hip.so}"
HIPCC="${HIPCC:-/opt/rocm/bin/hipcc}"
SRCS=(conv.hip conv_patch_bf16_a.hip conv_patch_bf16_b.hip conv_patch_f32_a.hip conv_patch_f32_b.hip conv_s2.hip conv_wide.hip conv_toep.hip heads.hip elementwise.hip norm_loss.hip optim_sn.hip metrics.hip input.hip)
HDRS=("$HERE"/*.h "$HERE/../../include/uegan_hip.h")
OBJS=()
mkdir -p "$HERE/_obj"
pids=()
for s in "${SRCS[@]}"; do
  o="$HERE/_obj/${s%.hip}.o"
  OBJS+=("$o")
  stale=0
  if [ ! -f "$o" ] || [ "$HERE/$s" -nt "$o" ]; then stale=1; fi
  for h in "${HDRS[@]}"; do if [ "$h" -nt "$o" ]; then stale=1; fi; done
  if [ "$stale" = 1 ]; then
    "$HIPCC" --offload-arch=gfx950 -O3 -std=c++17 -fPIC -munsafe-fp-atomics -Wall -Wno-unused-function -c "$HERE/$s" -o "$o" &
    pids+=($!)
  fi
done
rc=0
for p in "${pids[@]:-}"; do [ -n "$p" ] && { wait "$p" || rc=1; }; done
[ "$rc" = 0 ] || { echo "hipcc failed"; exit 1; }
"$HIPCC" --offload-arch=gfx950 -shared -fPIC "${OBJS[@]}" -o "$OUT"
echo "built $OUT"

#!/bin/bash
# Build libuegan_hip.so (16-bit storage format = bfloat16) and libuegan_hip_f16.so (the SAME sources with -DUEGAN_HALF_FP16: IEEE fp16) for
# gfx950 (MI355X).  hipcc cross-compiles without a GPU.  One object per translation unit, compiled in parallel; an object is rebuilt when
# its source or ANY header is newer.  usage: build.sh [out.so [f16-out.so]]   (UEGAN_BUILD_F16=0 skips the second library)
set -euo pipefail
HERE="$(cd "$(dirname "$0")" && pwd)"
OUT="${1:-$HERE/../libuegan_hip.so}"
OUT16="${2:-$(dirname "$OUT")/libuegan_hip_f16.so}"
HIPCC="${HIPCC:-/opt/rocm/bin/hipcc}"
SRCS=(conv.hip conv_stream_ex.hip conv_patch_bf16_a.hip conv_patch_bf16_b.hip conv_patch_f32_a.hip conv_patch_f32_b.hip conv_s2.hip conv_wide.hip conv_flat.hip conv_toep.hip heads.hip heads_mfma.hip elementwise.hip norm_loss.hip optim_sn.hip metrics.hip input.hip)
HDRS=("$HERE"/*.h "$HERE/../../include/uegan_hip.h")
build_one() {      # <object dir> <output .so> <extra flags...>
  local odir="$1" out="$2"; shift 2
  local objs=() pids=() rc=0
  mkdir -p "$odir"
  for s in "${SRCS[@]}"; do
    local o="$odir/${s%.hip}.o"
    objs+=("$o")
    local stale=0
    if [ ! -f "$o" ] || [ "$HERE/$s" -nt "$o" ]; then stale=1; fi
    for h in "${HDRS[@]}"; do if [ "$h" -nt "$o" ]; then stale=1; fi; done
    if [ "$stale" = 1 ]; then
      "$HIPCC" --offload-arch=gfx950 -O3 -std=c++17 -fPIC -munsafe-fp-atomics -Wall -Wno-unused-function "$@" -c "$HERE/$s" -o "$o" &
      pids+=($!)
    fi
  done
  for p in "${pids[@]:-}"; do [ -n "$p" ] && { wait "$p" || rc=1; }; done
  [ "$rc" = 0 ] || { echo "hipcc failed"; return 1; }
  "$HIPCC" --offload-arch=gfx950 -shared -fPIC "${objs[@]}" -o "$out"
  echo "built $out"
}
build_one "$HERE/_obj" "$OUT"
if [ "${UEGAN_BUILD_F16:-1}" != 0 ]; then build_one "$HERE/_obj_f16" "$OUT16" -DUEGAN_HALF_FP16; fi

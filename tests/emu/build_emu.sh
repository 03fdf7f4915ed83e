#!/bin/bash
# TEST INFRASTRUCTURE: compile the unmodified kernel sources for the host against tests/emu/hip/hip_runtime.h
# (fiber-based HIP emulator) -> tests/emu/_build/libuegan_emu.so (16-bit storage = bf16) and, the same sources with -DUEGAN_HALF_FP16,
# tests/emu/_build/libuegan_emu_f16.so (16-bit storage = fp16).  Never shipped, never loaded by the product.
set -euo pipefail
HERE="$(cd "$(dirname "$0")" && pwd)"
ROOT="$HERE/../.."
CXX="${EMU_CXX:-/opt/rocm/lib/llvm/bin/clang++}"
HDRS=("$ROOT"/uegan_amd/csrc/*.h "$HERE/hip/hip_runtime.h" "$ROOT/include/uegan_hip.h")
build_one() {      # <object dir> <output .so> <extra flags...>
  local OUT="$1" LIB="$2"; shift 2
  mkdir -p "$OUT"
  local pids=() OBJS=() rc=0
  for s in conv conv_stream_ex conv_patch_bf16_a conv_patch_bf16_b conv_patch_f32_a conv_patch_f32_b conv_s2 conv_wide conv_flat conv_toep heads heads_mfma elementwise norm_loss optim_sn metrics input; do
    local o="$OUT/$s.o"
    OBJS+=("$o")
    local src="$ROOT/uegan_amd/csrc/$s.hip"
    local stale=0
    if [ ! -f "$o" ] || [ "$src" -nt "$o" ]; then stale=1; fi
    for h in "${HDRS[@]}"; do if [ "$h" -nt "$o" ]; then stale=1; fi; done
    if [ "$stale" = 1 ]; then
      "$CXX" -x c++ -std=c++17 -O2 -g -fPIC -pthread -I"$HERE" -Wno-unused-function -Wno-reserved-identifier "$@" -c "$src" -o "$o" &
      pids+=($!)
    fi
  done
  for p in "${pids[@]:-}"; do [ -n "$p" ] && { wait "$p" || rc=1; }; done
  [ "$rc" = 0 ] || { echo "emulator compile failed"; return 1; }
  # link only when an object is newer than the library, into a temporary name + rename: several pytest-xdist workers run this script at once
  # and one of them may be loading the library
  local relink=0
  if [ ! -f "$LIB" ]; then relink=1; fi
  for o in "${OBJS[@]}"; do if [ "$o" -nt "$LIB" ]; then relink=1; fi; done
  if [ "$relink" = 1 ]; then
    "$CXX" -shared -fPIC -pthread "${OBJS[@]}" -o "$LIB.tmp.$$" && mv -f "$LIB.tmp.$$" "$LIB"
  fi
  echo "built $LIB"
}
build_one "$HERE/_build" "$HERE/_build/libuegan_emu.so"
build_one "$HERE/_build/f16" "$HERE/_build/libuegan_emu_f16.so" -DUEGAN_HALF_FP16

#!/bin/bash
# TEST INFRASTRUCTURE: compile the unmodified kernel sources for the host against tests/emu/hip/hip_runtime.h
# (fiber-based HIP emulator) -> tests/emu/_build/libuegan_emu.so.  Never shipped, never loaded by the product.
set -euo pipefail
HERE="$(cd "$(dirname "$0")" && pwd)"
ROOT="$HERE/../.."
CXX="${EMU_CXX:-/opt/rocm/lib/llvm/bin/clang++}"
OUT="$HERE/_build"
mkdir -p "$OUT"
pids=()
OBJS=()
HDRS=("$ROOT"/uegan_amd/csrc/*.h "$HERE/hip/hip_runtime.h" "$ROOT/include/uegan_hip.h")
for s in conv conv_patch_bf16_a conv_patch_bf16_b conv_patch_f32_a conv_patch_f32_b conv_s2 conv_wide conv_toep heads elementwise norm_loss optim_sn metrics input; do
  o="$OUT/$s.o"
  OBJS+=("$o")
  src="$ROOT/uegan_amd/csrc/$s.hip"
  stale=0
  if [ ! -f "$o" ] || [ "$src" -nt "$o" ]; then stale=1; fi
  for h in "${HDRS[@]}"; do if [ "$h" -nt "$o" ]; then stale=1; fi; done
  if [ "$stale" = 1 ]; then
    "$CXX" -x c++ -std=c++17 -O2 -g -fPIC -pthread -I"$HERE" -Wno-unused-function -Wno-reserved-identifier -c "$src" -o "$o" &
    pids+=($!)
  fi
done
rc=0
for p in "${pids[@]:-}"; do [ -n "$p" ] && { wait "$p" || rc=1; }; done
[ "$rc" = 0 ] || { echo "emulator compile failed"; exit 1; }
"$CXX" -shared -fPIC -pthread "${OBJS[@]}" -o "$OUT/libuegan_emu.so"
echo "built $OUT/libuegan_emu.so"

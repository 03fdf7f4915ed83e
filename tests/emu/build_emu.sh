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
for s in conv heads elementwise norm_loss optim_sn metrics; do
  o="$OUT/$s.o"
  OBJS+=("$o")
  src="$ROOT/uegan_amd/csrc/$s.hip"
  if [ ! -f "$o" ] || [ "$src" -nt "$o" ] || [ "$ROOT/uegan_amd/csrc/common.h" -nt "$o" ] || [ "$ROOT/uegan_amd/csrc/conv_internal.h" -nt "$o" ] || [ "$ROOT/uegan_amd/csrc/wgrad_tr.h" -nt "$o" ] || [ "$ROOT/uegan_amd/csrc/conv_stream.h" -nt "$o" ] || [ "$HERE/hip/hip_runtime.h" -nt "$o" ] || [ "$ROOT/include/uegan_hip.h" -nt "$o" ]; then
    "$CXX" -x c++ -std=c++17 -O2 -g -fPIC -pthread -I"$HERE" -Wno-unused-function -Wno-reserved-identifier -c "$src" -o "$o" &
    pids+=($!)
  fi
done
for p in "${pids[@]:-}"; do [ -n "$p" ] && wait "$p"; done
"$CXX" -shared -fPIC -pthread "${OBJS[@]}" -o "$OUT/libuegan_emu.so"
echo "built $OUT/libuegan_emu.so"

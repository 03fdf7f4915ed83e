#!/bin/bash
# Tools-only build of the kernel library with the timing ablations compiled in (-DUEGAN_TOOLS_BUILD): tools/_build/libuegan_hip_tools.so.
# The product library (uegan_amd/libuegan_hip.so) never contains them; tools/bench_conv.py --abl / --wide-abl load this one instead.
set -euo pipefail
HERE="$(cd "$(dirname "$0")" && pwd)"
SRC="$HERE/../uegan_amd/csrc"
OUT="$HERE/_build"
mkdir -p "$OUT/obj"
HIPCC="${HIPCC:-/opt/rocm/bin/hipcc}"
pids=()
for f in "$SRC"/*.hip; do
  o="$OUT/obj/$(basename "${f%.hip}").o"
  "$HIPCC" --offload-arch=gfx950 -O3 -std=c++17 -fPIC -munsafe-fp-atomics -DUEGAN_TOOLS_BUILD -Wno-unused-function -c "$f" -o "$o" &
  pids+=($!)
done
for p in "${pids[@]}"; do wait "$p"; done
"$HIPCC" --offload-arch=gfx950 -shared -fPIC "$OUT"/obj/*.o -o "$OUT/libuegan_hip_tools.so"
echo "built $OUT/libuegan_hip_tools.so"

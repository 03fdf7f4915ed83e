#!/bin/bash
for W in 192 0; do echo "== UEGAN_WIDE=$W"; UEGAN_WIDE=$W python tools/bench_wide.py 2>&1 | grep -v amdgpu; done
echo "== zero-filled, wide"; python tools/bench_wide.py --zero --cs 512 --bs 16 2>&1 | grep -v amdgpu
echo "== ABL 4"; UEGAN_WIDE_ABL=4 python tools/bench_wide.py --cs 64,512 --bs 16 2>&1 | grep -v amdgpu

#!/bin/bash
# usage: tools/gpu_ab_conv.sh <libA.so> "<bench_conv args>" : tools/bench_conv.py with the in-tree library, then with libA, on one box
export TMPDIR=/tmp
cp uegan_amd/libuegan_hip.so /tmp/lib_head.so
echo "== HEAD"; timeout 600 python tools/bench_conv.py $2 2>&1 | grep -E "^(G|D|VGG)\.|TOTAL" | cut -c1-100
cp $1 uegan_amd/libuegan_hip.so
echo "== $1"; timeout 600 python tools/bench_conv.py $2 2>&1 | grep -E "^(G|D|VGG)\.|TOTAL" | cut -c1-100
cp /tmp/lib_head.so uegan_amd/libuegan_hip.so

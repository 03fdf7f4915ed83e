#!/bin/bash
# conv_tall_kernel's rows-per-wave variants: tests, per-layer conv bench and whole-step A/B (UEGAN_TUNE_TALL_RPW = 5) on one box -> gpurun_out/
mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 1200 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider -k "tall or wide_kernel or uninitialised or at_size or fp16_storage" > gpurun_out/pytest_rpw.log 2>&1; echo "pytest rc=$?" >> gpurun_out/pytest_rpw.log
tail -8 gpurun_out/pytest_rpw.log | cut -c1-300
timeout 600 python tools/sweep_tuning.py "5=4" > gpurun_out/sweep_rpw.log 2>&1; cat gpurun_out/sweep_rpw.log | tail -6
timeout 600 python tools/bench_conv.py > gpurun_out/bench_conv_rpw2.log 2>&1
timeout 600 python tools/bench_conv.py --tune 5=4 > gpurun_out/bench_conv_rpw4.log 2>&1
paste <(grep -E "tall" gpurun_out/bench_conv_rpw2.log | cut -c1-110) <(grep -E "tall" gpurun_out/bench_conv_rpw4.log | awk '{print $(NF-3), $(NF-2), $(NF-1), $NF}') | head -60

export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_ops.py -m gpu -q -x -p no:cacheprovider -k "resample_pool" 2>&1 | tail -1
timeout 300 python tools/bench_elem.py 2>&1 | grep -E "upsample|maxpool|act_bwd"

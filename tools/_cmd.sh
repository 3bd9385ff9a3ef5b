export TMPDIR=/tmp; mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_hip_prefill.py -m gpu -q -x -p no:cacheprovider 2>&1 | tail -3
timeout 300 python tools/prefill_bench.py --chunks 512 --loop 1 2>&1 | grep "prefill n"
timeout 300 python tools/long_prefill_bench.py 2>&1 | tail -4

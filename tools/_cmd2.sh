#!/bin/bash
mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 300 python tools/prefill_bench.py --chunks 256,512 --loop 2 2>&1 | tail -3 | head -2
timeout 900 python -m pytest -q -p no:cacheprovider tests/test_hip_prefill.py -m gpu -x -q 2>&1 | tail -2

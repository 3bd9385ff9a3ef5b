#!/bin/bash
mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 1500 python -m pytest -q -p no:cacheprovider tests/test_hip_lazy.py tests/test_hip_runner.py tests/test_hip_fused.py -m gpu -x -q 2>&1 | tail -12

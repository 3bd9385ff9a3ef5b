#!/bin/bash
# the whole GPU suite on a box: gpurun --timeout 3000 -- "bash tools/gpu_tests.sh" (log: gpurun_out/gpu_tests_r06d.log)
mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 2400 python -m pytest -q -p no:cacheprovider tests -m gpu --maxfail=10 -q > gpurun_out/gpu_tests_r06d.log 2>&1; echo "tests rc=$?"; tail -15 gpurun_out/gpu_tests_r06d.log

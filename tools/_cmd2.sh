#!/bin/bash
mkdir -p gpurun_out; export TMPDIR=/tmp
echo "== f16w"; timeout 300 python tools/prefill_bench.py --chunks 128,256,512 --loop 2 2>&1 | tail -4
echo "== int8"; CRABML_HIP_TEST_HOOKS=1 CRABML_HIP_GEMM_INT8=1 timeout 300 python tools/prefill_bench.py --chunks 512 --loop 2 2>&1 | tail -2
timeout 900 python -m pytest -q -p no:cacheprovider tests/test_hip_prefill.py tests/test_hip_flash_attention.py -m gpu -x -q 2>&1 | tail -3

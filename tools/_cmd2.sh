#!/bin/bash
mkdir -p gpurun_out; export TMPDIR=/tmp
for v in 0 8; do echo "== f16w variant $v"; CRABML_HIP_TEST_HOOKS=1 CRABML_HIP_F16W=$v timeout 300 python tools/prefill_bench.py --chunks 256,512 --loop 2 2>&1 | tail -3 | head -2; done
timeout 900 python -m pytest -q -p no:cacheprovider tests/test_hip_prefill.py tests/test_hip_flash_attention.py -m gpu -x -q 2>&1 | tail -3

#!/bin/bash
mkdir -p gpurun_out; export TMPDIR=/tmp
echo "== stage_times default"; timeout 300 python tools/stage_times.py --steps 32
echo "== stage_times qkv two rounds"; CRABML_HIP_TEST_HOOKS=1 CRABML_HIP_QKV_UPFRONT=0 timeout 300 python tools/stage_times.py --steps 32 | grep -E "qkv|sum"
timeout 900 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-parity-check --no-context --no-prefill --no-gemv-points 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print({k:d.get(k) for k in ['value','value_strict','ms_per_step']}, d['fused_entry_point']['tokens_per_s'], d.get('c3_positions_0_127'))"

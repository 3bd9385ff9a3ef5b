#!/bin/bash
mkdir -p gpurun_out; export TMPDIR=/tmp
for wt in Q4_K Q8_0 Q4_1; do
timeout 900 python bench.py --wtype $wt --steps 20 --warmup 5 --no-cpu-baseline --no-context --no-prefill --no-gemv-points > gpurun_out/r06_bench_$wt.json 2>gpurun_out/r06_bench_$wt.err
python - <<PY
import json
d=json.load(open('gpurun_out/r06_bench_$wt.json'))
print('$wt', {k:d.get(k) for k in ['value','value_strict']}, d['fused_entry_point']['tokens_per_s'], d.get('c3_positions_0_127',{}).get('tokens_per_s'), d['roofline']['avg_launch_us'], d['roofline']['frac'])
PY
done

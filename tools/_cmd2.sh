#!/bin/bash
mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 1500 python bench.py --steps 20 --warmup 5 > gpurun_out/r06_bench_steps20_warmup5.json 2> gpurun_out/r06_bench.err; echo "rc=$?"
python - <<'PY'
import json
d=json.load(open('gpurun_out/r06_bench_steps20_warmup5.json'))
for k in ['value','value_strict','value_strict_through_reference_api','ms_per_step']: print(k, d.get(k))
print('fused', d['fused_entry_point']['tokens_per_s'])
print('c3', d.get('c3_positions_0_127'))
print('context', d.get('context'))
print('prefill', d.get('prefill'))
r=d['roofline']; print({k:r.get(k) for k in ['frac','frac_rocprof','achieved','avg_launch_us','traffic','frac_of_measured','measured_read_ceiling']})
print('cpu', d.get('cpu_baseline'))
print('trait', d.get('trait_path'))
pc=d.get('parity_check',{}); print({k:pc.get(k) for k in ['strict_bit_identical','fast_max_rel_logit_err','strict_tokens_per_s']}); 
for m in pc.get('models',[]): print(m.get('model')[:40], m.get('fast_over_reference_spread'), m.get('reference_api_equals_fused_entry_point_bitwise'), m.get('reference_api_tokens_served_by_fused_step'), m.get('fast_tokens_equal'), m.get('reference_avx2_vs_scalar_tokens_equal'))
print('gemv_points', [(p, v.get('frac_of_peak')) for p,v in d.get('gemv_points',{}).get('points',{}).items()])
PY

#!/bin/bash
# the bench lines of a round (driver arguments, defaults) and the per-format table: gpurun --timeout 3000 -- "bash tools/gpu_bench_lines.sh" (writes gpurun_out/r06_bench_*.json)
mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 1500 python bench.py --steps 20 --warmup 5 > gpurun_out/r06_bench_steps20_warmup5.json 2> gpurun_out/r06_bench.err; echo "rc=$?"
timeout 1500 python bench.py > gpurun_out/r06_bench_default.json 2> gpurun_out/r06_bench_default.err; echo "rc=$?"
for wt in Q4_K Q8_0 Q4_1 Q4_K_M Q6_K; do
timeout 900 python bench.py --wtype $wt --steps 20 --warmup 5 --no-cpu-baseline --no-context --no-gemv-points > gpurun_out/r06_bench_$wt.json 2>gpurun_out/r06_bench_$wt.err; echo "$wt rc=$?"
done
python - <<'PY'
import json
rows=[]
for wt,f in [('Q4_0','r06_bench_steps20_warmup5'),('Q4_K','r06_bench_Q4_K'),('Q8_0','r06_bench_Q8_0'),('Q4_1','r06_bench_Q4_1'),('Q4_K_M','r06_bench_Q4_K_M'),('Q6_K','r06_bench_Q6_K')]:
    d=json.load(open(f'gpurun_out/{f}.json'))
    r=d['roofline']
    rows.append(f"| {wt} | {d.get('value')} | {d['fused_entry_point']['tokens_per_s']} | {d.get('value_strict')} | {d.get('c3_positions_0_127',{}).get('tokens_per_s')} | {d.get('trait_path',{}).get('per_op_launches_tokens_per_s')} | {(d.get('prefill') or {}).get('prompt_tokens_per_s')} | {r['kernel'][:40]} | {r['avg_launch_us']} | {r['frac']} |")
print("\n".join(rows))
d=json.load(open('gpurun_out/r06_bench_steps20_warmup5.json'))
r=d['roofline']; print({k:r.get(k) for k in ['frac','frac_rocprof','traffic','avg_launch_us','rocprof_avg_launch_us','frac_of_measured']})
print(d['value'], d['fused_entry_point']['tokens_per_s'], d['value_strict'], d['cpu_baseline']['sample'], d['prefill']['prompt_tokens_per_s'], d['trait_path']['per_op_launches_tokens_per_s'])
d=json.load(open('gpurun_out/r06_bench_default.json')); print('default', d['value'], d['fused_entry_point']['tokens_per_s'], d['value_strict'], d['roofline']['frac'])
PY

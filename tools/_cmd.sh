export TMPDIR=/tmp; mkdir -p gpurun_out
timeout 300 ./build/attn_lab > gpurun_out/attn_lab_v11.log 2>&1; grep "STAGED" gpurun_out/attn_lab_v11.log | grep -v inside | head -12
timeout 1500 python -m pytest tests/test_hip_fused.py tests/test_hip_runner.py tests/test_real_fixture.py tests/test_hip_prefill.py -m gpu -q -x -p no:cacheprovider > gpurun_out/attn_tests.log 2>&1; echo "tests rc=$?"; tail -4 gpurun_out/attn_tests.log
timeout 300 python tools/ctx_attn_profile.py 1024 4096 8000 2>&1 | tail -3
timeout 300 python bench.py --steps 48 --repeats 3 --no-cpu-baseline --no-prefill > gpurun_out/bench_attn.json 2> gpurun_out/bench_attn.err; python - <<PY
import json
d=json.load(open("gpurun_out/bench_attn.json"))
print(d["value"], d["context"]["tokens_per_s_at_position"], d["parity_check"]["strict_bit_identical"], d["parity_check"]["fast_max_rel_logit_err"])
PY

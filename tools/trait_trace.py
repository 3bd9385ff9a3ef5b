"""The reference's unchanged runner (Llama2Runner<HipTensor>: one Tensor call after the other) decoding N tokens -- the workload
of a rocprofv3 --kernel-trace pass that shows which kernels its calls turn into (tools/gpu_profile.sh)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import crabml_amd as ca  # noqa: E402
from crabml_amd import synth  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 24
mode = sys.argv[2] if len(sys.argv) > 2 else "lazy"
dev = ca.HipTensorDevice(0, False, 0, False, mode)
model = synth.build_model(synth.SHAPES["llama3-8b"], synth.Q4_0, seed=8)
conf, w = synth.to_hip(model, dev)
r = ca.Llama2Runner(conf, w, dev, n + 16, True)
ids, sec, _ = r.timed_decode(1, n)
print({"mode": mode, "tokens": n, "tokens_per_s_under_the_tracer": round(n / sec, 1), "queue": dev.lazy_stats()})

"""Decode attention at long context: per-stage HIP-event times of the long-context kernels (stages 7 / 8 / 9 = scores /
softmax / pv) at a few cache fills, Llama-3-8B layer shape (8 layers: the per-layer kernels are the same), f16 KV cache.
usage: python tools/ctx_attn_profile.py [positions ...]"""
import sys

import numpy as np

sys.path.insert(0, ".")
import crabml_amd as ca  # noqa: E402
from crabml_amd import synth  # noqa: E402

POS = [int(a) for a in sys.argv[1:]] or [256, 1024, 4096, 8000]
L = 8
model = synth.build_model(synth.SHAPES["llama3-8b"], synth.Q4_0, seed=3, n_layers=L)
dev = ca.HipTensorDevice(0)
conf, w = synth.to_hip(model, dev)
rng = np.random.default_rng(0)
NAMES = {1: "k_qkv", 2: "wo", 3: "gateup", 4: "down", 5: "classifier", 6: "norm", 7: "attn/scores", 8: "softmax", 9: "pv"}
for p in POS:
    r = ca.HipLlamaRunner(conf, w, dev, p + 64, True, False)  # eager: events around every stage
    toks = [int(t) for t in rng.integers(1, 1000, size=p)]
    r.prefill(toks)
    r.forward(5, p)
    dev.sync()
    dev.prof_enable(True)
    n = 8
    for i in range(n):
        r.forward(7 + i, p + 1 + i)
    recs = dev.prof_read()
    dev.prof_enable(False)
    line = {NAMES.get(x["stage"], str(x["stage"])): round(x["kernel_ms"] * 1e3 / x["launches"], 2) for x in recs if x["launches"]}
    g = ca.HipLlamaRunner(conf, w, dev, p + 64, True)
    g.prefill(toks)
    g.decode_greedy(5, 4)
    import time
    dev.sync(); t0 = time.perf_counter(); g.decode_greedy(5, 32); dev.sync(); dt = (time.perf_counter() - t0) / 32
    print(f"pos {p}: us per launch {line}  | graph step {dt*1e6:.0f} us for {L} layers = {dt*1e6/L:.1f} us/layer (+classifier)", flush=True)
    del r, g

"""Decode attention at long context: per-stage HIP-event times of the long-context kernels at a few cache fills, Llama-3-8B
layer shape (8 layers: the per-layer kernels are the same), f16 KV cache.  Stages: the fast step's default k_attn_flash (7) +
k_attn_flash_merge (8); with FLAGS=4194304 (EXACT_ATTENTION) the exact kernels scores (7) / softmax (8) / pv (9); with
FLAGS=2097152 (FLASH_TICKET) the single-launch flash form (7).
usage: [FLAGS=n] python tools/ctx_attn_profile.py [positions ...]"""
import os
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
FLAGS = int(os.environ.get("FLAGS", "0"))
NAMES = {1: "k_qkv", 2: "wo", 3: "gateup", 4: "down", 5: "classifier", 6: "norm", 7: "attn/scores", 8: "softmax", 9: "pv"}
for p in POS:
    r = ca.HipLlamaRunner(conf, w, dev, p + 64, True, False, extra_flags=FLAGS)  # eager: events around every stage
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
    g = ca.HipLlamaRunner(conf, w, dev, p + 64, True, extra_flags=FLAGS)
    g.prefill(toks)
    g.decode_greedy(5, 4)
    import time
    dev.sync(); t0 = time.perf_counter(); g.decode_greedy(5, 32); dev.sync(); dt = (time.perf_counter() - t0) / 32
    print(f"flags {FLAGS} pos {p}: us per launch {line}  | graph step {dt*1e6:.0f} us for {L} layers = {dt*1e6/L:.1f} us/layer (+classifier)", flush=True)
    del r, g

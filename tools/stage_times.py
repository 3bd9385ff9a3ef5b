#!/usr/bin/env python3
"""Per-stage kernel times of the fused decode step (eager replay + dispatch-timestamp events).
usage: stage_times.py [--no-prefetch] [--steps N] [--pos0 P]"""
import argparse
import sys

sys.path.insert(0, ".")
import crabml_amd as ca
from crabml_amd import synth

ap = argparse.ArgumentParser()
ap.add_argument("--no-prefetch", action="store_true")
ap.add_argument("--no-norm-epilogue", action="store_true")
ap.add_argument("--flags", type=int, default=0)
ap.add_argument("--hidden", type=int, default=0, help="experiment: override the ffn hidden size")
ap.add_argument("--layers", type=int, default=0)
ap.add_argument("--vocab", type=int, default=0, help="experiment: override the vocabulary size")
ap.add_argument("--attn-long-from", type=int, default=0)
ap.add_argument("--steps", type=int, default=16)
ap.add_argument("--pos0", type=int, default=8)
ap.add_argument("--burn", type=float, default=0.0, help="seconds of back-to-back decoding before the measured steps (thermal / power state)")
ap.add_argument("--model", default="llama3-8b")
ap.add_argument("--wtype", default="Q4_0")
a = ap.parse_args()
NAMES = {1: "qkv", 2: "wo+res", 3: "gateup_q", 4: "down+res", 5: "classifier", 6: "norm_quant", 7: "attn|scores", 8: "attn softmax", 9: "attn pv", 10: "ffn (k_ffn)"}
shape = synth.SHAPES[a.model]
if a.hidden:
    shape = synth.ModelShape(**{**shape.__dict__, "hidden": a.hidden})
if a.vocab:
    shape = synth.ModelShape(**{**shape.__dict__, "vocab": a.vocab})
model = synth.build_model(shape, synth.TYPE_BY_NAME[a.wtype], seed=8, n_layers=a.layers or None)
dev = ca.HipTensorDevice(0)
conf, w = synth.to_hip(model, dev)
r = ca.HipLlamaRunner(conf, w, dev, a.pos0 + a.steps + 8, True, False, not a.no_prefetch, norm_epilogue=not a.no_norm_epilogue, extra_flags=a.flags, attn_long_from=a.attn_long_from)
if a.burn > 0:
    import time
    g = ca.HipLlamaRunner(conf, w, dev, 256, True)
    t0 = time.time()
    while time.time() - t0 < a.burn:
        g.reset()
        g.decode_greedy(1, 200)
    dev.sync()
    del g
r.decode_greedy(1, a.pos0)
dev.sync()
dev.prof_enable(True)
r.decode_greedy(1, a.steps)
recs = dev.prof_read()
dev.prof_enable(False)
tot = 0.0
for x in sorted(recs, key=lambda x: x["stage"]):
    us = x["kernel_ms"] * 1e3 / x["launches"]
    per_tok = x["kernel_ms"] * 1e3 / a.steps
    tot += per_tok
    gb = x["algo_bytes"] / (x["kernel_ms"] * 1e-3) / 1e9 if x["algo_bytes"] else 0
    print(f"{NAMES.get(x['stage'], x['stage']):12s} {x['launches'] / a.steps:6.1f}/tok  avg {us:7.2f} us  {per_tok:8.1f} us/tok  {gb:7.0f} GB/s")
print(f"sum of kernel time {tot:.1f} us/token (positions {a.pos0}..{a.pos0 + a.steps - 1})")

#!/usr/bin/env python3
"""Token-id level counterpart of `crabml-cli generate` for the hip backend: load a llama GGUF file through the C++
loader (crabml_amd/csrc/host/gguf.hpp), prefill a prompt in batched passes, decode greedily on the device.
The tokenizer and the samplers other than greedy stay on the reference's side of the boundary (out of scope here),
so the prompt is given as token ids.

usage: generate.py model.gguf [--prompt 1,15043,3186] [--steps 64] [--seq-len N] [--f32-kv] [--strict]
       generate.py --synth tiny-gqa:Q4_K_M   (writes a synthetic file to a temp dir first: a self-contained demo)"""
import argparse
import os
import sys
import tempfile
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import crabml_amd as ca
from crabml_amd import synth

ap = argparse.ArgumentParser()
ap.add_argument("gguf", nargs="?")
ap.add_argument("--synth", default=None, help="SHAPE:TYPE, e.g. tiny-gqa:Q4_0, llama3-8b:Q4_K_M")
ap.add_argument("--prompt", default="1,365,400,282,7,9,11,13")
ap.add_argument("--steps", type=int, default=32)
ap.add_argument("--seq-len", type=int, default=0)
ap.add_argument("--f32-kv", action="store_true")
ap.add_argument("--strict", action="store_true", help="strict-order device: the reference's scalar summation order, bit for bit")
a = ap.parse_args()

path = a.gguf
tmp = None
if a.synth:
    shape_name, typ = a.synth.split(":")
    k_m = typ.upper() == "Q4_K_M"
    model = synth.build_model(synth.SHAPES[shape_name], synth.Q4_K if k_m else synth.TYPE_BY_NAME[typ], seed=8, k_m_mix=k_m)
    tmp = tempfile.TemporaryDirectory()
    path = os.path.join(tmp.name, "model.gguf")
    synth.write_gguf(model, path)
if not path:
    ap.error("give a GGUF file or --synth SHAPE:TYPE")

t0 = time.perf_counter()
gf = ca.GGUFFile(path)
conf = gf.load_config()
types = sorted({t[2] for t in gf.tensor_infos()})
print(f"{path}: GGUF v{gf.version}, {len(gf.tensor_infos())} tensors (ggml types {types}), arch {gf.architecture}, "
      f"dim {conf.embedding_dim}, layers {conf.n_layers}, heads {conf.n_heads}/{conf.n_kv_heads}, vocab {conf.vocab_size}")
dev = ca.HipTensorDevice(0, False, 0, a.strict)
weights = gf.load_weights(conf, dev)
dev.sync()
print(f"loaded + uploaded in {time.perf_counter() - t0:.2f} s")
prompt = [int(t) for t in a.prompt.split(",") if t]
seq_len = a.seq_len or min(conf.seq_len, len(prompt) + a.steps + 8)
r = ca.HipLlamaRunner(conf, weights, dev, seq_len, not a.f32_kv)
t0 = time.perf_counter()
logits = r.prefill(prompt)
t_prefill = time.perf_counter() - t0
first = int(len(logits) - 1 - logits[::-1].argmax())  # the LAST maximum (sampler.rs:109-116)
t0 = time.perf_counter()
ids = [first] + [int(t) for t in r.decode_greedy(first, a.steps - 1)] if a.steps > 1 else [first]
t_decode = time.perf_counter() - t0
print(f"prefill: {len(prompt)} tokens in {t_prefill * 1e3:.2f} ms ({len(prompt) / t_prefill:.0f} tok/s)")
if a.steps > 1:
    print(f"decode:  {a.steps - 1} tokens in {t_decode * 1e3:.2f} ms ({(a.steps - 1) / t_decode:.1f} tok/s)")
print("tokens:", ",".join(str(t) for t in ids))

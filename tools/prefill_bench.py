#!/usr/bin/env python3
"""Prompt-processing throughput of the batched prefill (crabml_hip_llama_prefill) against the token loop.
usage: prefill_bench.py [--model llama3-8b] [--wtype Q4_0] [--n 512] [--chunks 64,128,256,512] [--layers N]
Prints one line per chunk size: prompt tokens/s (host clock around the blocking call, 2nd of 2 runs) and the int8
MFMA rate the weight GEMMs reach (2 * m * k ops per prompt row and weight element)."""
import argparse
import sys
import time

sys.path.insert(0, ".")
import crabml_amd as ca
from crabml_amd import synth

ap = argparse.ArgumentParser()
ap.add_argument("--model", default="llama3-8b")
ap.add_argument("--wtype", default="Q4_0")
ap.add_argument("--n", type=int, default=512)
ap.add_argument("--chunks", default="64,128,256,512")
ap.add_argument("--layers", type=int, default=0)
ap.add_argument("--loop", type=int, default=64, help="tokens of the token-loop baseline")
ap.add_argument("--flags", type=int, default=0, help="extra CRABML_HIP_LLAMA_* flags (A/B runs)")
a = ap.parse_args()
shape = synth.SHAPES[a.model]
k_m = a.wtype.upper() == "Q4_K_M"  # llama.cpp's mix: Q4_K body, attn_v / ffn_down in Q6_K on some layers, Q6_K classifier
model = synth.build_model(shape, synth.Q4_K if k_m else synth.TYPE_BY_NAME[a.wtype], seed=8, n_layers=a.layers or None, k_m_mix=k_m)
dev = ca.HipTensorDevice(0)
conf, w = synth.to_hip(model, dev)
L = a.layers or shape.n_layers
hd = shape.dim // shape.n_heads
ops_per_row = 2.0 * L * (shape.dim * shape.dim * 2 + 2 * shape.dim * hd * shape.n_kv_heads + 3 * shape.dim * shape.hidden)
toks = [(7 * i + 1) % shape.vocab for i in range(a.n)]
for chunk in [int(c) for c in a.chunks.split(",")]:
    r = ca.HipLlamaRunner(conf, w, dev, a.n + 8, True, prefill_chunk=chunk, extra_flags=a.flags)
    best = None
    for rep in range(2):
        r.reset()
        dev.sync()
        t0 = time.perf_counter()
        r.prefill(toks)
        dt = time.perf_counter() - t0
        best = dt if best is None else min(best, dt)
    print(f"prefill n={a.n} chunk={chunk:4d}: {best * 1e3:8.2f} ms  {a.n / best:9.0f} prompt tok/s  "
          f"{ops_per_row * a.n / best / 1e12:6.1f} TOP/s in the weight GEMMs", flush=True)
    del r
r = ca.HipLlamaRunner(conf, w, dev, a.n + 8, True)
r.forward(toks[0], 0)
dev.sync()
t0 = time.perf_counter()
for i in range(1, a.loop):
    r.forward_async(toks[i], i)
dev.sync()
dt = time.perf_counter() - t0
print(f"token loop ({a.loop - 1} forwards): {dt / (a.loop - 1) * 1e3:.3f} ms/token  {(a.loop - 1) / dt:.0f} prompt tok/s")

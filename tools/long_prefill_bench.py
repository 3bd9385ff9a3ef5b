#!/usr/bin/env python3
"""Long-prompt prefill: time of an n-token prompt (8B shape by default): the fast step's flash attention on the f16 matrix cores
(default), the exact kernels (EXACT_ATTENTION) with and without the row-tiled PV pass.
usage: long_prefill_bench.py [--n 4096] [--model llama3-8b]"""
import argparse
import sys
import time

sys.path.insert(0, ".")
import crabml_amd as ca
from crabml_amd import synth

ap = argparse.ArgumentParser()
ap.add_argument("--model", default="llama3-8b")
ap.add_argument("--wtype", default="Q4_0")
ap.add_argument("--n", type=int, default=4096)
ap.add_argument("--layers", type=int, default=0)
ap.add_argument("--only-default", action="store_true")
a = ap.parse_args()
shape = synth.SHAPES[a.model]
model = synth.build_model(shape, synth.TYPE_BY_NAME[a.wtype], seed=8, n_layers=a.layers or None)
dev = ca.HipTensorDevice(0)
conf, w = synth.to_hip(model, dev)
toks = [(7 * i + 1) % shape.vocab for i in range(a.n)]
for label, flags in (("flash attention on the f16 matrix cores (default fast step)", 0), ("exact kernels, row-tiled PV (EXACT_ATTENTION)", 4194304),
                     ("exact kernels, PV one prompt row per workgroup (+ flag 16384)", 4194304 + 16384)):
    if a.only_default and flags:
        continue
    r = ca.HipLlamaRunner(conf, w, dev, a.n + 8, True, extra_flags=flags)
    best = None
    for rep in range(2):
        r.reset()
        dev.sync()
        t0 = time.perf_counter()
        r.prefill(toks)
        dt = time.perf_counter() - t0
        best = dt if best is None else min(best, dt)
    print(f"prefill n={a.n} {label}: {best * 1e3:8.2f} ms  {a.n / best:9.0f} prompt tok/s", flush=True)
    del r

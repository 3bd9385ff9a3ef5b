#!/usr/bin/env python3
"""A/B of the engine (crabml_amd/csrc/engine.hpp) against the 5-launch layer on ONE model upload: for each configuration
(flags, consumer waves, ring depth, thinned loader) a fresh HipLlamaRunner decodes W warm-up + R x K timed greedy steps under the
hipGraph (tokens/s, median region), then 16 eager steps with the dispatch-timestamp events give the per-stage kernel times.
usage: engine_ab.py [--model llama3-8b] [--layers N] [--steps 48] [--configs base,eng,eng:nc=7,eng:d=5,eng:thin=1,...]"""
import argparse
import json
import os
import sys
import time

sys.path.insert(0, ".")
import numpy as np

import crabml_amd as ca
from crabml_amd import synth

ap = argparse.ArgumentParser()
ap.add_argument("--model", default="llama3-8b")
ap.add_argument("--wtype", default="Q4_0")
ap.add_argument("--layers", type=int, default=0)
ap.add_argument("--steps", type=int, default=48)
ap.add_argument("--warmup", type=int, default=8)
ap.add_argument("--repeats", type=int, default=3)
ap.add_argument("--configs", default="base,eng,eng:nc=7,eng:nc=5,eng:thin=1,eng:d=5,eng:noprefetch")
ap.add_argument("--check", action="store_true", help="compare every configuration's logits with the first one's (bit identity)")
a = ap.parse_args()
ENGINE = 524288
NAMES = {1: "qkv", 2: "wo", 3: "gateup", 4: "down", 5: "cls", 6: "norm", 7: "attn", 8: "softmax", 9: "pv", 10: "k_ffn", 11: "engine"}
model = synth.build_model(synth.SHAPES[a.model], synth.TYPE_BY_NAME[a.wtype], seed=8, n_layers=a.layers or None)
dev = ca.HipTensorDevice(0)
conf, w = synth.to_hip(model, dev)
seq = a.warmup + a.steps + 40
ref = None
for cfg in a.configs.split(","):
    parts = cfg.split(":")
    flags = ENGINE if parts[0] == "eng" else 2097152 if parts[0] == "tail" else int(parts[0][5:]) if parts[0].startswith("flags") else 0
    env = {}
    prefetch = True
    for p in parts[1:]:
        if p == "noprefetch":
            prefetch = False
            continue
        k, v = p.split("=")
        env[{"nc": "CRABML_HIP_ENGINE_NC", "d": "CRABML_HIP_ENGINE_D", "thin": "CRABML_HIP_ENGINE_THIN", "lag": "CRABML_HIP_ENGINE_LAG"}[k]] = v
    for k in ("CRABML_HIP_ENGINE_NC", "CRABML_HIP_ENGINE_D", "CRABML_HIP_ENGINE_THIN", "CRABML_HIP_ENGINE_LAG"):
        os.environ.pop(k, None)
    os.environ.update(env)
    try:
        r = ca.HipLlamaRunner(conf, w, dev, seq, True, True, prefetch, extra_flags=flags)
        rates = []
        for rep in range(a.repeats):
            r.reset()
            tok = int(r.decode_greedy(1, a.warmup)[-1])
            dev.sync()
            t0 = time.perf_counter()
            r.decode_greedy(tok, a.steps)
            dev.sync()
            rates.append(a.steps / (time.perf_counter() - t0))
        rates.sort()
        line = {"config": cfg, "tok_s": round(rates[len(rates) // 2], 1), "all": [round(x, 1) for x in rates]}
        if a.check:
            r.reset()
            lg = [r.forward(t, i).copy() for i, t in enumerate([1, 365, 400, 282, 7, 9])]
            if ref is None:
                ref = lg
            line["bit_identical_to_first"] = all(np.array_equal(x.view(np.uint32), y.view(np.uint32)) for x, y in zip(lg, ref))
        del r
        e = ca.HipLlamaRunner(conf, w, dev, 64, True, False, prefetch, extra_flags=flags)
        e.decode_greedy(1, a.warmup)
        dev.sync()
        dev.prof_enable(True)
        e.decode_greedy(1, 16)
        recs = dev.prof_read()
        dev.prof_enable(False)
        line["stage_us"] = {NAMES.get(x["stage"], str(x["stage"])): round(x["kernel_ms"] * 1e3 / x["launches"], 2) for x in sorted(recs, key=lambda x: x["stage"])}
        line["kernel_us_per_token"] = round(sum(x["kernel_ms"] for x in recs) * 1e3 / 16, 1)
        del e
    except Exception as ex:  # a configuration that fails must not take the others down
        line = {"config": cfg, "error": repr(ex)}
    print(json.dumps(line), flush=True)

#!/usr/bin/env python3
"""Where the engine launch (crabml_amd/csrc/engine.hpp) spends its time: s_memrealtime stamps (100 MHz) written by every
workgroup at its phase boundaries, plus accumulated wait / busy times of the loader and of the first three consumer waves.
Prints, per phase, min / median / max over the workgroups in microseconds relative to the earliest workgroup start, for the
layers asked for (the stamps of the LAST decode step stay in the buffer).
usage: CRABML_HIP_ENGINE_STAMPS=1 engine_stamps.py [--layers 4] [--steps 12] [--show-layers 1,2] [env knobs as for engine_ab.py]"""
import argparse
import os
import sys

os.environ["CRABML_HIP_ENGINE_STAMPS"] = "1"
sys.path.insert(0, ".")
import numpy as np

import crabml_amd as ca
from crabml_amd import synth

ap = argparse.ArgumentParser()
ap.add_argument("--model", default="llama3-8b")
ap.add_argument("--layers", type=int, default=4)
ap.add_argument("--steps", type=int, default=12)
ap.add_argument("--show-layers", default="1,2")
ap.add_argument("--graph", type=int, default=1)
a = ap.parse_args()
model = synth.build_model(synth.SHAPES[a.model], synth.Q4_0, seed=8, n_layers=a.layers)
dev = ca.HipTensorDevice(0)
conf, w = synth.to_hip(model, dev)
r = ca.HipLlamaRunner(conf, w, dev, a.steps + 16, True, bool(a.graph), True, extra_flags=524288)
r.decode_greedy(1, a.steps)
dev.sync()
s = model.shape
G = min(256, max(s.dim // 16, s.hidden // 32))
W = 64
PH = [(0, "edge wave starts"), (1, "attention planes staged"), (2, "own wo slots done"), (3, "wo edge: hop 1 done, chunk published"),
      (4, "normalized x gathered (phase 2)"), (5, "own gate/up slots done"), (6, "h gathered (phase 3)"), (7, "own down slots done"),
      (8, "down edge done (end)"), (16, "loader starts"), (17, "loader: slot 0 issued"), (20, "loader: wo issued"),
      (21, "loader: gate/up issued"), (22, "loader: down issued"), (23, "loader: all landed")]
for l in [int(x) for x in a.show_layers.split(",")]:
    st = r.engine_stamps(l, G * W).reshape(G, W).astype(np.float64)
    t0 = st[:, 16][st[:, 16] > 0].min()
    print(f"== layer {l}: {G} workgroups; times in us since the first loader start")
    for i, name in PH:
        v = st[:, i]
        v = (v[v > 0] - t0) / 100.0
        if v.size:
            print(f"  {name:42s} n={v.size:3d}  min {v.min():7.2f}  med {np.median(v):7.2f}  max {v.max():7.2f}")
    for i, name in ((24, "loader: ring-full stall (sum)"), (25, "loader: vmcnt waits (sum)")):
        v = st[:, i] / 100.0
        print(f"  {name:42s} n={G:3d}  min {v.min():7.2f}  med {np.median(v):7.2f}  max {v.max():7.2f}")
    for op, oname in enumerate(("wo", "gate/up", "down")):
        for cw in range(3):
            b = 32 + (op * 3 + cw) * 3
            n = st[:, b + 2]
            m = n > 0
            if m.any():
                print(f"  consumer {cw} {oname:8s}: slots med {np.median(n[m]):4.0f}  wait-for-slot med {np.median(st[m, b]) / 100:6.2f} us"
                      f"  compute med {np.median(st[m, b + 1]) / 100:6.2f} us  = {np.median(st[m, b + 1] / n[m]) / 100:5.2f} us/slot")
    heavy = st[: max(1, s.hidden // 32 - G), :] if s.hidden // 32 > G else None
    if heavy is not None:
        print(f"  (workgroups 0..{heavy.shape[0] - 1} carry two gate/up blocks: own gate/up done med {(np.median(heavy[:, 5]) - t0) / 100:.2f} us,"
              f" the others {(np.median(st[heavy.shape[0]:, 5]) - t0) / 100:.2f} us)")

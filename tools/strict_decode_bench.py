#!/usr/bin/env python3
"""Decode rate of the STRICT-ORDER device (CRABML_HIP_FLAG_STRICT_ORDER: every sum in the reference's scalar order, logits bit-identical
to the oracle) on a synthetic model.  usage: python tools/strict_decode_bench.py [--wtype Q4_0] [--layers N] [--steps 32]
Under rocprofv3 --kernel-trace it yields the per-kernel table of the strict step (tools/rocpd_summary.py)."""
import argparse
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import crabml_amd as ca  # noqa: E402
from crabml_amd import synth  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--model", default="llama3-8b")
ap.add_argument("--wtype", default="Q4_0")
ap.add_argument("--layers", type=int, default=None)
ap.add_argument("--steps", type=int, default=32)
ap.add_argument("--fast", action="store_true", help="the default (fast) device instead, for comparison")
ap.add_argument("--flags", type=int, default=0, help="extra CRABML_HIP_LLAMA_* flags (A/B runs)")
a = ap.parse_args()
k_m = a.wtype.upper() == "Q4_K_M"  # llama.cpp's mix: Q4_K body, attn_v / ffn_down in Q6_K on some layers, Q6_K classifier
model = synth.build_model(synth.SHAPES[a.model], synth.Q4_K if k_m else synth.TYPE_BY_NAME[a.wtype], seed=8, n_layers=a.layers, k_m_mix=k_m)
dev = ca.HipTensorDevice(0) if a.fast else ca.HipTensorDevice(0, False, 0, True)
conf, w = synth.to_hip(model, dev)
r = ca.HipLlamaRunner(conf, w, dev, a.steps + 24, True, extra_flags=a.flags)
r.decode_greedy(1, 8)
dev.sync()
t0 = time.perf_counter()
r.decode_greedy(1, a.steps)
dev.sync()
dt = time.perf_counter() - t0
print(f"{'fast' if a.fast else 'strict-order'} device, {a.model} {a.wtype}{'' if a.layers is None else f' ({a.layers} layers)'}: "
      f"{a.steps / dt:.1f} tok/s, {1e3 * dt / a.steps:.3f} ms per token")

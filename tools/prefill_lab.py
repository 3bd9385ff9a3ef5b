#!/usr/bin/env python3
"""Batched matmul_vec (prefill shape) on the MFMA skinny-GEMM path: time per call, weight GB/s, int8 TOP/s.
usage: prefill_lab.py [--wtype Q4_0]"""
import argparse
import sys
import time

import numpy as np

sys.path.insert(0, ".")
import crabml_amd as ca
from crabml_amd import synth

ap = argparse.ArgumentParser()
ap.add_argument("--wtype", default="Q4_0")
a = ap.parse_args()
typ = synth.TYPE_BY_NAME[a.wtype]
gt = {synth.Q4_0: ca.GGMLType.Q4_0, synth.Q8_0: ca.GGMLType.Q8_0}[typ]
dev = ca.HipTensorDevice(0)
rng = np.random.default_rng(1)
print(f"{'shape':>16} {'b':>4} {'us/call':>9} {'weight GB/s':>12} {'TOP/s (2mkb)':>13} {'tokens/s (this GEMM only)':>26}")
for (m, k) in [(4096, 4096), (14336, 4096), (4096, 14336), (128256, 4096)]:
    raw = synth.random_blocks(rng, m * k, typ)
    w = ca.HipTensor.from_cpu(raw, [m, k], gt, dev)
    wbytes = m * k // 32 * synth.BLOCK_BYTES[typ]
    for b in (1, 16, 32, 64, 128, 256):
        x = ca.HipTensor.new(rng.standard_normal(b * k).astype(np.float32), [b, k] if b > 1 else [k], dev)
        w.matmul_vec(x)  # quantizes the rhs (cached per buffer) + warms up
        dev.sync()
        n = 20 if m < 100000 else 5
        t0 = time.perf_counter()
        for _ in range(n):
            y = w.matmul_vec(x)
        dev.sync()
        us = (time.perf_counter() - t0) / n * 1e6
        print(f"{m:>8}x{k:<7} {b:>4} {us:>9.1f} {wbytes / us / 1e3:>12.1f} {2.0 * m * k * b / us / 1e6:>13.1f} {b / us * 1e6:>26.0f}")

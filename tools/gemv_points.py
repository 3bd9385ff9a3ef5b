#!/usr/bin/env python3
"""Single-kernel GEMV points of SURVEY.md 8(d): per weight format, the four Llama-3-8B shapes, each launch on a
DIFFERENT weight buffer (>= 600 MB of distinct buffers per shape, cycled: the 256 MB Infinity Cache never holds the
next one), 20 warm-up + 200 timed launches, kernel time from start/stop events on the launch's own stream:
median / p10 / p90 and algorithmic GB/s.  Also: the hipMemcpyDtoD rate of the box (read + write).
usage: gemv_points.py [--formats Q4_0,Q8_0,...]"""
import argparse
import sys

import numpy as np

sys.path.insert(0, ".")
import crabml_amd as ca
from crabml_amd import synth

ap = argparse.ArgumentParser()
ap.add_argument("--formats", default="Q4_0,Q8_0,Q4_1,Q4_K,Q6_K,Q8_K")
a = ap.parse_args()
GT = {synth.Q4_0: ca.GGMLType.Q4_0, synth.Q8_0: ca.GGMLType.Q8_0, synth.Q4_1: ca.GGMLType.Q4_1, synth.Q4_K: ca.GGMLType.Q4K,
      synth.Q6_K: ca.GGMLType.Q6K, synth.Q8_K: ca.GGMLType.Q8K}
dev = ca.HipTensorDevice(0)
rng = np.random.default_rng(3)
print("| format | shape (m x k) | algorithmic MB | median us | p10 us | p90 us | GB/s at median | % of 8 TB/s |")
print("|---|---|---:|---:|---:|---:|---:|---:|")
for fname in a.formats.split(","):
    typ = synth.TYPE_BY_NAME[fname]
    for (m, k) in [(4096, 4096), (14336, 4096), (4096, 14336), (128256, 4096)]:
        wbytes = m * k // synth.BLOCK_ELEMS[typ] * synth.BLOCK_BYTES[typ]
        algo = wbytes + 4 * k + 4 * m
        ncopies = max(2, -(-600_000_000 // wbytes))
        raw = synth.random_blocks(rng, m * k, typ)
        ws = []
        for c in range(ncopies):
            ws.append(ca.HipTensor.from_cpu(np.roll(raw, 4096 * c), [m, k], GT[typ], dev))
        x = ca.HipTensor.new(rng.standard_normal(k).astype(np.float32), [k], dev)
        for i in range(20):
            ws[i % ncopies].matmul_vec(x)
        dev.sync()
        dev.prof_enable(True)
        for i in range(200):
            ws[i % ncopies].matmul_vec(x)
        ms = dev.prof_read_launches()
        dev.prof_enable(False)
        us = np.sort(ms.astype(np.float64) * 1e3)
        med, p10, p90 = np.median(us), us[len(us) // 10], us[len(us) * 9 // 10]
        print(f"| {fname} | {m} x {k} | {algo / 1e6:.2f} | {med:.2f} | {p10:.2f} | {p90:.2f} | {algo / med / 1e3:.0f} | {algo / med / 1e3 / 80:.1f} |")
        del ws

# device-to-device copy rate (hipMemcpyAsync D2D via Tensor::dup): bytes read + bytes written
import time
n = 256 << 20  # 1 GiB of f32
t = ca.HipTensor.new(np.zeros(n, dtype=np.float32), [n], dev)
for _ in range(2):
    u = t.dup()
dev.sync()
t0 = time.perf_counter()
for _ in range(10):
    u = t.dup()
dev.sync()
dt = (time.perf_counter() - t0) / 10
print(f"\nhipMemcpy device-to-device, 1 GiB: {dt * 1e6:.0f} us per copy = {2 * n * 4 / dt / 1e9:.0f} GB/s (read + write)")

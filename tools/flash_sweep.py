"""Decode step time per layer vs cached positions for the attention forms of the fast step (Llama-3-8B layer shape, 8 layers,
hipGraph): the staged one-workgroup-per-head kernel (exact), the exact long-context kernels, k_attn_flash + merge launch,
k_attn_flash with the last-arriver merge, and the slice size (CRABML_HIP_FLASH_MIN_ROWS, a CRABML_HIP_TEST_HOOKS=1 tuning hook).
usage: python tools/flash_sweep.py > profiles/r04_flash_sweep.log"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, ".")
os.environ["CRABML_HIP_TEST_HOOKS"] = "1"
import crabml_amd as ca  # noqa: E402
from crabml_amd import synth  # noqa: E402

POS = [int(a) for a in sys.argv[1:]] or [32, 64, 96, 128, 160, 192, 224, 256, 512, 1024, 2048, 4096, 8000]
L = 8
EXACT, TICKET, NO_LONG = 4194304, 2097152, 64
model = synth.build_model(synth.SHAPES["llama3-8b"], synth.Q4_0, seed=3, n_layers=L)
dev = ca.HipTensorDevice(0)
conf, w = synth.to_hip(model, dev)
rng = np.random.default_rng(0)
FORMS = [("one-wg exact (k_attn_s)", EXACT | NO_LONG, None, 10 ** 9), ("exact long (3 kernels)", EXACT, None, 1),
         ("flash+merge rows>=128", 0, 128, 1), ("flash+merge rows>=256", 0, 256, 1), ("flash+merge rows>=64", 0, 64, 1),
         ("flash ticket rows>=128", TICKET, 128, 1), ("flash ticket rows>=256", TICKET, 256, 1)]
print("# us per layer (graph step / %d layers, classifier included), by cached positions" % L)
print("%-28s" % "form" + "".join("%8d" % p for p in POS))
for name, flags, rows, long_from in FORMS:
    if rows is not None:
        os.environ["CRABML_HIP_FLASH_MIN_ROWS"] = str(rows)
    line = []
    for p in POS:
        if name.startswith("one-wg") and p > 1024:
            line.append(float("nan"))
            continue
        g = ca.HipLlamaRunner(conf, w, dev, ((p + 72) // 8) * 8, True, extra_flags=flags, attn_long_from=long_from)
        g.prefill([int(t) for t in rng.integers(1, 1000, size=p)])
        g.decode_greedy(5, 4)
        dev.sync()
        t0 = time.perf_counter()
        g.decode_greedy(5, 32)
        dev.sync()
        line.append((time.perf_counter() - t0) / 32 * 1e6 / L)
        del g
    print("%-28s" % name + "".join("%8.1f" % v for v in line), flush=True)

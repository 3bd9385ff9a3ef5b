"""Does a smaller GRID (fewer slices per kv head, i.e. fewer workgroups that only read the position and leave) shorten the flash
pair at mid contexts?  us per layer (8 layers of the 8B shape, hipGraph) by cached positions, for grids of S slices per kv head.
usage: python tools/flash_slices_sweep.py"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, ".")
os.environ["CRABML_HIP_TEST_HOOKS"] = "1"
import crabml_amd as ca  # noqa: E402
from crabml_amd import synth  # noqa: E402

POS = [128, 256, 512, 1024, 2048, 4096]
L = 8
model = synth.build_model(synth.SHAPES["llama3-8b"], synth.Q4_0, seed=3, n_layers=L)
dev = ca.HipTensorDevice(0)
conf, w = synth.to_hip(model, dev)
rng = np.random.default_rng(0)
print("%-22s" % "slices in the grid" + "".join("%8d" % p for p in POS))
for S in (2, 4, 8, 16, 32):
    os.environ["CRABML_HIP_FLASH_SLICES"] = str(S)
    line = []
    for p in POS:
        g = ca.HipLlamaRunner(conf, w, dev, ((p + 72) // 8) * 8, True)
        g.prefill([int(t) for t in rng.integers(1, 1000, size=p)])
        g.decode_greedy(5, 4)
        dev.sync()
        t0 = time.perf_counter()
        g.decode_greedy(5, 32)
        dev.sync()
        line.append((time.perf_counter() - t0) / 32 * 1e6 / L)
        del g
    print("%-22s" % ("S = %d" % S) + "".join("%8.1f" % v for v in line), flush=True)

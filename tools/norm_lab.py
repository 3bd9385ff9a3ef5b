"""The fast step's hop-free norm (deferred 1 / rms) against the exact gather form: tokens/s and distance from the strict-order
device (= the oracle, bit for bit) on the benchmark model and on a zero-mean Q4_0 model, teacher-forced on the strict tokens."""
import argparse
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import crabml_amd as ca  # noqa: E402
from crabml_amd import synth  # noqa: E402

EXACT_NORM = 8388608
ap = argparse.ArgumentParser()
ap.add_argument("--model", default="llama3-8b")
ap.add_argument("--wtype", default="Q4_0")
ap.add_argument("--positions", type=int, default=12)
ap.add_argument("--layers", type=int, default=None)
args = ap.parse_args()

sdev = ca.HipTensorDevice(0, False, 0, True)
fdev = ca.HipTensorDevice(0)
out = []
for zero_mean in (False, True):
    model = synth.build_model(synth.SHAPES[args.model], synth.TYPE_BY_NAME[args.wtype], seed=8, n_layers=args.layers)
    if zero_mean:
        synth.flip_scale_signs(model)
    sconf, sw = synth.to_hip(model, sdev)
    fconf, fw = synth.to_hip(model, fdev)
    n = args.positions
    s = ca.HipLlamaRunner(sconf, sw, sdev, 256, True)
    ref, toks, tok = [], [], 1
    for i in range(n):
        toks.append(tok)
        lg = s.forward(tok, i)
        ref.append(lg.copy())
        tok = int(len(lg) - 1 - np.argmax(lg[::-1]))
    row = {"model": args.model, "wtype": args.wtype, "zero_mean": zero_mean}
    for name, flags in (("deferred", 0), ("exact_norm", EXACT_NORM)):
        f = ca.HipLlamaRunner(fconf, fw, fdev, 256, True, extra_flags=flags)
        errs, same = [], []
        for i in range(n):
            lg = f.forward(toks[i], i)
            errs.append(float(np.max(np.abs(lg - ref[i])) / np.max(np.abs(ref[i]))))
            same.append(int(len(lg) - 1 - np.argmax(lg[::-1])) == (toks[i + 1] if i + 1 < n else int(len(ref[i]) - 1 - np.argmax(ref[i][::-1]))))
        f.reset()
        f.decode_greedy(1, 8)
        fdev.sync()
        best = None
        for _ in range(3):
            f.reset()
            f.decode_greedy(1, 8)
            fdev.sync()
            t0 = time.perf_counter()
            f.decode_greedy(1, 64)
            fdev.sync()
            dt = time.perf_counter() - t0
            best = dt if best is None else min(best, dt)
        row[name] = {"tok_s": round(64 / best, 1), "err_median": float(np.median(errs)), "err_max": float(max(errs)),
                     "errs": [float("%.2e" % e) for e in errs], "greedy_equal": int(sum(same))}
        del f
    out.append(row)
    print(json.dumps(row), flush=True)
    del s, sw, fw

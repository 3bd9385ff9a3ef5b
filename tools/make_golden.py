#!/usr/bin/env python3
"""Regenerates tests/golden/oracle_decode.json: digests of the ORACLE's logits on seeded synthetic models.

The reference (Rust nightly) cannot be executed in this environment, so these are not reference outputs: the
reference's own known-answer values are transcribed, with file:line, in tests/test_oracle_kats.py.  This file pins
the oracle itself -- the parity anchor of every GPU test -- against silent drift between rounds: any change to
oracle/ that moves a single logit bit fails tests/test_oracle_golden.py and has to be justified against the
reference source.  usage: python tools/make_golden.py"""
import hashlib
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from crabml_amd import synth  # noqa: E402
from oracle import oracle as o  # noqa: E402
from tests.helpers import to_oracle  # noqa: E402

TOKENS = [1, 365, 400, 282, 7, 9]
CASES = [(shape, fmt, kv16) for shape in ("tiny-gqa",) for fmt in ("Q4_0", "Q8_0", "Q4_1", "Q5_0", "Q5_1", "Q2_K", "Q3_K", "Q4_K", "Q5_K", "Q6_K", "Q8_K", "F16", "F32")
         for kv16 in (True, False)] + [("15m", "Q4_0", True), ("15m", "Q8_0", False),
                                      # llama.cpp's Q4_K_M recipe: mixed GGML types inside a layer + Q6_K classifier
                                      ("tiny-gqa", "Q4_K_M", True), ("tiny-gqa", "Q4_K_M", False)]


def run(shape, fmt, kv16, avx2=False):
    k_m = fmt == "Q4_K_M"
    model = synth.build_model(synth.SHAPES[shape], synth.Q4_K if k_m else synth.TYPE_BY_NAME[fmt], seed=20250103,
                              n_layers=8 if k_m else 2, k_m_mix=k_m)
    odev = o.OracleDevice(thread_num=2, use_avx2=avx2)
    conf, w = to_oracle(model, odev)
    r = o.OracleLlamaRunner(conf, w, odev, 32, kv16)
    logits = [r.forward([t], i).copy() for i, t in enumerate(TOKENS)]
    raw = b"".join(x.tobytes() for x in logits)
    return {"sha256": hashlib.sha256(raw).hexdigest(), "first_logits_step0": [float(v) for v in logits[0][:4]],
            "argmax_last_step": int(o.argmax_last(logits[-1]))}


def main():
    out = {"tokens": TOKENS, "seed": 20250103, "n_layers": 2, "seq_len": 32, "cases": {}}
    for shape, fmt, kv16 in CASES:
        out["cases"][f"{shape}/{fmt}/{'f16kv' if kv16 else 'f32kv'}"] = run(shape, fmt, kv16)
    with open(os.path.join(ROOT, "tests", "golden", "oracle_decode.json"), "w") as f:
        json.dump(out, f, indent=1)
    print(f"wrote {len(out['cases'])} cases")


if __name__ == "__main__":
    main()

"""How far apart are the reference's OWN two CPU builds on the benchmark's models?

crabml has a scalar and an AVX2 order of the Q8_0 / Q4_0 dots (buf_q8_0.rs:228-286, buf_q4_0.rs:215-253); the oracle restates both.
They differ only in how the per-block terms are associated -- exactly the freedom the HIP fast path takes -- so their distance on a
model is the yardstick for the fast path's distance from the scalar oracle on that model (the rhs quantizer truncates: one ulp in
a GEMV output can move the largest element of a block between the levels 126 and 127, buf_q8_0.rs:119-124).

Slow (8B-shape models on the CPU): runs only with CRABML_RUN_SLOW=1; the output of the last run is committed as
profiles/r04_reference_order_sensitivity.log."""
import os

import numpy as np
import pytest

from crabml_amd import synth
from oracle import oracle as o
from tests.helpers import to_oracle

TOKS = [1, 365, 400, 282]


def spread(model):
    outs = []
    for avx2 in (False, True):
        odev = o.OracleDevice(thread_num=os.cpu_count() or 2, use_avx2=avx2)
        conf, w = to_oracle(model, odev)
        r = o.OracleLlamaRunner(conf, w, odev, 16, True)
        outs.append([r.forward([t], i).copy() for i, t in enumerate(TOKS)])
    errs = [float(np.max(np.abs(a - b)) / np.max(np.abs(a))) for a, b in zip(*outs)]
    same = [bool(o.argmax_last(a) == o.argmax_last(b)) for a, b in zip(*outs)]
    return errs, same


def flip_scale_signs(model, rng):
    """Q4_0 blocks with d of either sign (what a quantizer that divides by the signed maximum produces, buf_q4_0.rs:96-104): the
    synthetic blocks of synth.random_blocks all carry d > 0, i.e. a mean level of -0.5 d -- a common-mode component in every GEMV"""
    for name, t in model.tensors.items():
        if t.typ == synth.Q4_0:
            blk = t.data.reshape(-1, 18)
            blk[:, 1] ^= (rng.integers(0, 2, size=blk.shape[0], dtype=np.uint8) << 7)


@pytest.mark.skipif(not os.environ.get("CRABML_RUN_SLOW"), reason="8B-shape models on the CPU: minutes; set CRABML_RUN_SLOW=1")
@pytest.mark.parametrize("shape,layers", [("tiny-gqa", 2), ("llama3-8b", 32)])
def test_scalar_vs_avx2_order_of_the_reference(shape, layers):
    rows = []
    for fmt, signs in (("Q8_0", False), ("Q4_0", False), ("Q4_0", True)):
        model = synth.build_model(synth.SHAPES[shape], synth.TYPE_BY_NAME[fmt], seed=8, n_layers=layers)
        if signs:
            flip_scale_signs(model, np.random.default_rng(5))
        errs, same = spread(model)
        rows.append(f"{shape} x {layers} layers, {fmt}{' with d of either sign' if signs else ''}: scalar vs AVX2 order, max|d| / max|logit| per "
                    f"position {[f'{e:.1e}' for e in errs]}, greedy token equal {same}")
        print(rows[-1], flush=True)
    assert len(rows) == 3

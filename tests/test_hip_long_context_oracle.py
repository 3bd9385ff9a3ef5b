"""Strict decode against the oracle PAST 1024 cached positions (round-2 verdict: the fused step's softmax switched to a block
tree beyond 1024 positions on the strict device as well, and nothing past position 299 had been compared with the oracle).

softmax.rs:43-48 sums the exponentials in one scalar loop at any length; the strict-order device now keeps that order at every
length in all three softmax hosts (k_attn, k_attn_s, k_attn_softmax), so the logits at positions 1023 / 1024 / 1025 / 2047 /
2048 / 4095 are asserted BIT-IDENTICAL to the oracle's token loop (llama2.rs:184-281, batch_matmul.rs:28-99, softmax.rs:36-54),
through the multi-workgroup long-context kernels and through the one-workgroup-per-head kernel.  The fast device is held to a
pinned tolerance at the same positions, with the oracle's own tokens, in both of its long-context forms: the exact kernels
(CRABML_HIP_LLAMA_EXACT_ATTENTION: the reference's f16 PV chain, block-tree row sum beyond 1024 positions -> FAST_TOL) and the
default split-KV kernel with f32 accumulation (k_attn_flash, from 224 positions -> FLASH_TOL, round 4).  Q4_K weights (a
round-to-nearest rhs quantizer: nothing but the attention deviation survives) measure that deviation by itself; Q4_0 shows it
next to the truncating quantizer's +-1 flips."""
import numpy as np
import pytest

from crabml_amd import synth
from oracle import oracle as o
from tests.helpers import check_fast, to_oracle

EXACT = 4194304  # CRABML_HIP_LLAMA_EXACT_ATTENTION

pytestmark = pytest.mark.gpu
CHECK = (0, 223, 224, 1023, 1024, 1025, 2047, 2048, 4095)


@pytest.mark.parametrize("fmt", ["Q4_0", "Q4_K"])
def test_strict_and_fast_decode_equal_the_oracle_up_to_position_4095(ca, fmt):
    s = synth.SHAPES["tiny-gqa"]
    model = synth.build_model(s, synth.TYPE_BY_NAME[fmt], seed=41, n_layers=1)
    rng = np.random.default_rng(5)
    n = CHECK[-1] + 1
    toks = [int(t) for t in rng.integers(0, s.vocab, size=n)]
    odev = o.OracleDevice(thread_num=8)
    oconf, ow = to_oracle(model, odev)
    orr = o.OracleLlamaRunner(oconf, ow, odev, n + 8, True)
    sdev = ca.HipTensorDevice(0, False, 0, True)
    sconf, sw = synth.to_hip(model, sdev)
    strict_long = ca.HipLlamaRunner(sconf, sw, sdev, n + 8, True)                   # long-context kernels from 224 positions
    strict_one = ca.HipLlamaRunner(sconf, sw, sdev, n + 8, True, extra_flags=64)    # NO_LONG_ATTENTION: one workgroup per head
    fdev = ca.HipTensorDevice(0)
    fconf, fw = synth.to_hip(model, fdev)
    fast = ca.HipLlamaRunner(fconf, fw, fdev, n + 8, True, extra_flags=EXACT)
    flash = ca.HipLlamaRunner(fconf, fw, fdev, n + 8, True)  # k_attn_flash from 224 cached positions
    errs, ferrs = [], []
    for i, t in enumerate(toks):
        ref = orr.forward([t], i)
        if i in CHECK:
            for name, r in (("long", strict_long), ("one-wg", strict_one)):
                got = r.forward(t, i)
                assert np.array_equal(got.view(np.uint32), ref.view(np.uint32)), f"strict {name} position {i}"
            got = fast.forward(t, i)
            errs.append(float(np.max(np.abs(got - ref)) / np.max(np.abs(ref))))
            got = flash.forward(t, i)
            ferrs.append(float(np.max(np.abs(got - ref)) / np.max(np.abs(ref))))
        else:
            strict_long.forward_async(t, i)
            strict_one.forward_async(t, i)
            fast.forward_async(t, i)
            flash.forward_async(t, i)
    check_fast(f"fused/tiny-gqa-1layer/long-context/{fmt}", fmt, np.array(errs))
    check_fast(f"fused/tiny-gqa-1layer/long-context-flash/{fmt}", "FLASH:" + fmt, np.array(ferrs))

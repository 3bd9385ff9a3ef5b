"""GPU parity: on-the-fly activation quantization is BYTE-identical to the reference quantizers
(buf_q8_0.rs:87-134 truncation, buf_q8_1.rs:90-129, buf_q8_k.rs:84-131 round-half-away)."""
import numpy as np
import pytest

from oracle import oracle as o

pytestmark = pytest.mark.gpu


def cases(n, seed):
    rng = np.random.default_rng(seed)
    yield "normal", rng.standard_normal(n).astype(np.float32)
    yield "wide", (rng.standard_normal(n) * 10.0 ** rng.integers(-6, 5, n)).astype(np.float32)
    x = rng.standard_normal(n).astype(np.float32)
    x[: n // 4] = 0.0  # leading all-zero blocks: 0/0 -> NaN paths (q=0 for Q8_0, q=-128 for Q8_1, d=0 for Q8_K)
    yield "zero_blocks", x
    x = rng.integers(-8, 8, n).astype(np.float32)  # many exact ties (|x| equal, opposite signs) + exact .5 products
    yield "ties", x
    x = rng.standard_normal(n).astype(np.float32)
    x[::7] *= -1.0
    x[3::256] = -np.abs(x).max() * 2  # negative max element
    yield "neg_max", x
    yield "tiny", (rng.standard_normal(n) * 1e-30).astype(np.float32)
    yield "huge", (rng.standard_normal(n) * 1e30).astype(np.float32)


@pytest.mark.parametrize("qname", ["Q8_0", "Q8_1", "Q8K"])
@pytest.mark.parametrize("n", [256, 288 * 8, 4096, 14336])
def test_quantize_bytes_identical(ca, hdev, qname, n):
    if qname == "Q8K" and n % 256:
        n = (n // 256) * 256
    otyp = {"Q8_0": o.Q8_0, "Q8_1": o.Q8_1, "Q8K": o.Q8_K}[qname]
    for name, x in cases(n, n + len(qname)):
        got = ca.HipTensor.new(x, [n], hdev).debug_quantize(getattr(ca.GGMLType, qname))
        ref = o.quantize(x, otyp)
        assert got.shape == ref.shape
        if not np.array_equal(got, ref):
            bb = o.BLOCK_BYTES[otyp]
            bad = np.nonzero((got.reshape(-1, bb) != ref.reshape(-1, bb)).any(axis=1))[0]
            raise AssertionError(f"{qname} n={n} case={name}: {bad.size} blocks differ, first {bad[:4]}")

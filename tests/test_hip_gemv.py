"""GPU parity for Tensor::matmul_vec on every weight format of the hot path.

Gates, strongest first:
  1. integer level, BIT-EXACT: per 32-element group  sum(unpacked_w * q8)  through the kernel's own
     nibble-unpack / v_dot4 code  ==  the reference's scalar loops  (north_star: "bit-exactly at the
     integer unpack level").
  2. dequantized rows (copy_rows_from on a quantized table), BIT-EXACT.
  3. GEMV outputs: |hip - oracle| <= GEMV_REL * sum_i |w_i x_i|  -- the f32 re-association bound; the
     oracle is the reference's scalar-fallback order, and its AVX2 order is checked to the same bound.
"""
import numpy as np
import pytest

from crabml_amd import synth
from oracle import oracle as o
from tests.helpers import GEMV_REL, gemv_order_bound

pytestmark = pytest.mark.gpu

FORMATS = ["Q4_0", "Q8_0", "Q4_1", "Q5_0", "Q5_1", "Q2_K", "Q3_K", "Q4_K", "Q5_K", "Q6_K", "Q8_K"]
HT = {"Q4_0": "Q4_0", "Q8_0": "Q8_0", "Q4_1": "Q4_1", "Q5_0": "Q5_0", "Q5_1": "Q5_1", "Q2_K": "Q2K", "Q3_K": "Q3K", "Q4_K": "Q4K", "Q5_K": "Q5K", "Q6_K": "Q6K", "Q8_K": "Q8K", "F32": "F32", "F16": "F16"}


def make(fmt, m, k, seed):
    typ = synth.TYPE_BY_NAME[fmt]
    rng = np.random.default_rng(seed)
    raw = synth.random_blocks(rng, m * k, typ)
    x = rng.standard_normal(k).astype(np.float32)
    return typ, raw, x


# shapes: 15M model (9 / 24 blocks per row: not a multiple of 64 lanes), ragged m (odd, < R), 8B layer shapes
SHAPES_32 = [(288, 288), (768, 288), (288, 768), (1, 32), (3, 64), (5, 2080), (1000, 4096), (4096, 4096),
             (1024, 4096), (300, 14336)]
SHAPES_256 = [(3, 256), (5, 768), (512, 512), (1000, 4096), (257, 14336), (1024, 1024)]


def shapes_for(fmt):
    return SHAPES_256 if fmt in ("Q2_K", "Q3_K", "Q4_K", "Q5_K", "Q6_K", "Q8_K") else SHAPES_32


@pytest.mark.parametrize("fmt", FORMATS)
def test_block_dots_bit_exact(ca, hdev, fmt):
    for (m, k) in shapes_for(fmt)[:6]:
        typ, raw, x = make(fmt, m, k, m * 7 + k)
        w = ca.HipTensor.from_cpu(raw, [m, k], getattr(ca.GGMLType, HT[fmt]), hdev)
        hx = ca.HipTensor.new(x, [k], hdev)
        xq = o.quantize(x, o.rhs_dtype(typ))
        rb = o.BLOCK_BYTES[typ] * (k // o.BLOCK_ELEMS[typ])
        for row in sorted({0, m // 2, m - 1}):
            got = w.debug_block_dots(row, hx)
            ref = o.block_dots(raw[row * rb:(row + 1) * rb], typ, xq, k)
            assert np.array_equal(got, ref), f"{fmt} ({m},{k}) row {row}"


@pytest.mark.parametrize("fmt", FORMATS + ["F32", "F16"])
def test_dequant_rows_bit_exact(ca, hdev, odev, fmt):
    m, k = 37, 512
    typ, raw, _ = make(fmt, m, k, 99)
    w = ca.HipTensor.from_cpu(raw, [m, k], getattr(ca.GGMLType, HT[fmt]), hdev)
    ow = o.OracleTensor.from_bytes(raw, typ, [m, k], odev)
    rows = [36, 0, 17]
    dst = ca.HipTensor.alloc([3, k], ca.GGMLType.F32, hdev)
    dst.copy_rows_from(w, rows)
    odst = o.OracleTensor.alloc([3, k], o.F32, odev)
    odst.copy_rows_from(ow, rows)
    assert np.array_equal(dst.export().view(np.uint32), odst.export().view(np.uint32))


@pytest.mark.parametrize("fmt", FORMATS)
def test_gemv_vs_oracle(ca, hdev, odev, fmt):
    for (m, k) in shapes_for(fmt):
        typ, raw, x = make(fmt, m, k, m * 13 + k)
        w = ca.HipTensor.from_cpu(raw, [m, k], getattr(ca.GGMLType, HT[fmt]), hdev)
        got = w.matmul_vec(ca.HipTensor.new(x, [k], hdev))
        assert got.shape() == [m]
        got = got.export()
        ref = o.OracleTensor.from_bytes(raw, typ, [m, k], odev).matmul_vec(o.OracleTensor.new(x, [k], odev)).export()
        bound = gemv_order_bound(raw, typ, x, m, k) * GEMV_REL * (8 if fmt in ("Q4_1", "Q5_1", "Q2_K", "Q4_K", "Q5_K") else 1) + 1e-30
        err = np.abs(got.astype(np.float64) - ref.astype(np.float64))
        assert np.all(err <= bound), f"{fmt} ({m},{k}): max err/bound {np.max(err / bound):.3f}"
        if typ == o.Q4_K:
            xq = o.quantize(x, o.Q8_K)
            rb = 144 * (k // 256)
            assert o.q4k_overflow_count(raw[:rb], xq, k) >= 0  # informational: i16 hazard of buf_q4_k.rs:240


def test_gemv_avx2_order_is_within_the_same_bound(odev):
    """The reference has two CPU accumulation orders (scalar fallback / AVX2 when k % 1024 == 0,
    buf_q4_0.rs:220-223); they differ from each other by the same re-association bound we allow the GPU."""
    if not o.lib().co_have_avx2():
        pytest.skip("no avx2")
    m, k = 64, 4096
    typ, raw, x = make("Q4_0", m, k, 5)
    xq = o.quantize(x, o.Q8_0)
    rb = 18 * (k // 32)
    a = np.array([o.vec_dot(raw[r * rb:(r + 1) * rb], typ, xq, k, avx2=False) for r in range(m)])
    b = np.array([o.vec_dot(raw[r * rb:(r + 1) * rb], typ, xq, k, avx2=True) for r in range(m)])
    assert np.all(np.abs(a - b) <= gemv_order_bound(raw, typ, x, m, k) * GEMV_REL)


def test_gemv_f32_f16_weights(ca, hdev, odev):
    for fmt, (m, k) in [("F32", (64, 64)), ("F32", (172, 64)), ("F32", (512, 172)), ("F16", (48, 96))]:
        typ, raw, x = make(fmt, m, k, m + k)
        w = ca.HipTensor.from_cpu(raw, [m, k], getattr(ca.GGMLType, HT[fmt]), hdev)
        got = w.matmul_vec(ca.HipTensor.new(x, [k], hdev)).export()
        ref = o.OracleTensor.from_bytes(raw, typ, [m, k], odev).matmul_vec(o.OracleTensor.new(x, [k], odev)).export()
        bound = gemv_order_bound(raw, typ, x, m, k) * GEMV_REL + 1e-30
        assert np.all(np.abs(got.astype(np.float64) - ref) <= bound)


def test_gemv_batched_rhs_and_quant_cache(ca, hdev, odev):
    """(m,k) @ (b,k) -> (b,m); and the per-buffer activation-quantization cache must be invalidated by
    every in-place write (q/k/v share one quantization of x, then x changes)."""
    m, k, b = 96, 1024, 3
    typ, raw, _ = make("Q4_0", m, k, 1)
    rng = np.random.default_rng(2)
    x = rng.standard_normal(b * k).astype(np.float32)
    w = ca.HipTensor.from_cpu(raw, [m, k], ca.GGMLType.Q4_0, hdev)
    ow = o.OracleTensor.from_bytes(raw, typ, [m, k], odev)
    hx = ca.HipTensor.new(x, [b, k], hdev)
    got = w.matmul_vec(hx)
    assert got.shape() == [b, m]
    ref = ow.matmul_vec(o.OracleTensor.new(x, [b, k], odev)).export()
    assert np.allclose(got.export(), ref, rtol=1e-4, atol=1e-4)
    got2 = w.matmul_vec(hx).export()  # cached quantization: identical result
    assert np.array_equal(got.export(), got2)
    hx = hx.scale_inplace(0.5)  # in-place write must invalidate the cache
    got3 = w.matmul_vec(hx).export()
    ref3 = ow.matmul_vec(o.OracleTensor.new(x * np.float32(0.5), [b, k], odev)).export()
    assert np.allclose(got3, ref3, rtol=1e-4, atol=1e-4)
    assert not np.array_equal(got3, got2)


@pytest.mark.parametrize("fmt", ["Q4_0", "Q8_0", "Q4_1", "Q8_K"])
def test_batched_rhs_on_the_matrix_cores_is_bit_exact(ca, hdev, odev, fmt):
    """b >= 16 activation rows take the MFMA skinny-GEMM path (gemm_mfma.hip): exact integer tiles from
    v_mfma_i32_16x16x32_i8, scaled block by block like the reference's scalar loop -- every output equals the
    scalar-order oracle BIT FOR BIT (stronger than the single-row fast GEMV, which re-associates).  Ragged shapes:
    m not a multiple of 16, b not a multiple of 16 / 64, k with 1 and 9 blocks."""
    for (m, k, b) in [(16, 32, 16), (37, 288, 17), (100, 1024, 40), (288, 288, 64), (64, 2048, 100), (1000, 4096, 33)]:
        if fmt == "Q8_K":
            k = max(256, k // 256 * 256)  # super-blocks of 256
        typ, raw, _ = make(fmt, m, k, m + k + b)
        rng = np.random.default_rng(b)
        x = rng.standard_normal(b * k).astype(np.float32)
        w = ca.HipTensor.from_cpu(raw, [m, k], getattr(ca.GGMLType, HT[fmt]), hdev)
        got = w.matmul_vec(ca.HipTensor.new(x, [b, k], hdev))
        assert got.shape() == [b, m]
        ref = o.OracleTensor.from_bytes(raw, typ, [m, k], odev).matmul_vec(o.OracleTensor.new(x, [b, k], odev)).export()
        assert np.array_equal(got.export().view(np.uint32), ref.view(np.uint32)), f"{fmt} m={m} k={k} b={b}"


def test_batched_rhs_q4_k_on_the_matrix_cores(ca, hdev, odev):
    """Q4_K weights x >= 16 Q8_K activation rows (the prompt of a *_K_M file): integer MFMA tiles per 32-element
    sub-block, folded with the 6-bit sub-block scales in integers, the minimum term as two more MFMAs per super-block
    (gemm_mfma.hip: k_gemm_mfma_q4k).  Same integers as the reference, f32 accumulation per super-block: within the
    GEMV re-association bound of the oracle (x8 for Q4_K, as for the single-row kernel).  Ragged m / b, 1 .. 16
    super-blocks, both column-tile widths."""
    for (m, k, b) in [(16, 256, 16), (37, 512, 17), (100, 1024, 40), (256, 256, 64), (64, 2048, 100), (1000, 4096, 33),
                      (4096, 512, 64)]:
        typ, raw, _ = make("Q4_K", m, k, m + k + b)
        rng = np.random.default_rng(b)
        x = rng.standard_normal(b * k).astype(np.float32)
        w = ca.HipTensor.from_cpu(raw, [m, k], ca.GGMLType.Q4K, hdev)
        got = w.matmul_vec(ca.HipTensor.new(x, [b, k], hdev)).export().reshape(b, m)
        ref = o.OracleTensor.from_bytes(raw, typ, [m, k], odev).matmul_vec(o.OracleTensor.new(x, [b, k], odev)).export().reshape(b, m)
        for r in range(b):
            bound = gemv_order_bound(raw, typ, x[r * k:(r + 1) * k], m, k) * GEMV_REL * 8 + 1e-30
            assert np.all(np.abs(got[r] - ref[r]) <= bound), f"m={m} k={k} b={b} row {r}: {np.max(np.abs(got[r] - ref[r]) / bound)}"
        # and equal, within the same bound, to the single-row kernel fed the same rows
        one = w.matmul_vec(ca.HipTensor.new(x[:k].copy(), [k], hdev)).export()
        assert np.all(np.abs(got[0] - one) <= gemv_order_bound(raw, typ, x[:k], m, k) * GEMV_REL * 8 + 1e-30)


def test_batched_rhs_q6_k_on_the_matrix_cores(ca, hdev, odev):
    """Q6_K weights x >= 16 Q8_K rows (k_gemm_mfma_q6k): every 32-element unit is multiplied twice with half of the weight
    operand zeroed (a Q6_K scale covers 16 elements), 6-bit values rebuilt from ql + qh, int8 group scales folded in
    integers, the -32 offset as an MFMA over (scales x quant sums).  Within the GEMV bound of the oracle."""
    for (m, k, b) in [(16, 256, 16), (37, 512, 17), (100, 1024, 40), (256, 256, 64), (64, 2048, 100), (1000, 4096, 33)]:
        typ, raw, _ = make("Q6_K", m, k, m + k + b)
        rng = np.random.default_rng(b)
        x = rng.standard_normal(b * k).astype(np.float32)
        w = ca.HipTensor.from_cpu(raw, [m, k], ca.GGMLType.Q6K, hdev)
        got = w.matmul_vec(ca.HipTensor.new(x, [b, k], hdev)).export().reshape(b, m)
        ref = o.OracleTensor.from_bytes(raw, typ, [m, k], odev).matmul_vec(o.OracleTensor.new(x, [b, k], odev)).export().reshape(b, m)
        for r in range(b):
            bound = gemv_order_bound(raw, typ, x[r * k:(r + 1) * k], m, k) * GEMV_REL * 8 + 1e-30
            assert np.all(np.abs(got[r] - ref[r]) <= bound), f"m={m} k={k} b={b} row {r}: {np.max(np.abs(got[r] - ref[r]) / bound)}"


def test_gemv_errors(ca, hdev):
    typ, raw, x = make("Q4_0", 8, 64, 3)
    w = ca.HipTensor.from_cpu(raw, [8, 64], ca.GGMLType.Q4_0, hdev)
    with pytest.raises(ca.CrabmlError):  # inner dims differ (matmul_vec.rs:19)
        w.matmul_vec(ca.HipTensor.new(np.zeros(32, dtype=np.float32), [32], hdev))
    with pytest.raises(ca.CrabmlError):  # not a block multiple
        ca.HipTensor.from_cpu(raw[:18 * 3], [3, 30], ca.GGMLType.Q4_0, hdev)
    with pytest.raises(ca.CrabmlError):  # too few bytes for the shape
        ca.HipTensor.from_cpu(raw[:18], [8, 64], ca.GGMLType.Q4_0, hdev)
    with pytest.raises(ca.CrabmlError):  # non-contiguous rhs
        w.matmul_vec(ca.HipTensor.new(np.zeros(128, dtype=np.float32), [64, 2], hdev).transpose([1, 0]))


def test_linearity_at_full_8b_shape(ca, hdev):
    """Size-independent property at BASELINE.json's full shape (the CPU oracle would take minutes on
    the whole matrix): with activations that quantize exactly (small integers * power of two),
    W.(x1 + x2) == W.x1 + W.x2 up to f32 rounding, and rows sampled against the oracle."""
    m, k = 14336, 4096
    typ, raw, _ = make("Q4_0", m, k, 42)
    w = ca.HipTensor.from_cpu(raw, [m, k], ca.GGMLType.Q4_0, hdev)
    rng = np.random.default_rng(1)
    # each 32-block holds +-127 so d = 1.0 exactly and q = x exactly for integer x in [-127,127]
    def ints():
        v = rng.integers(-60, 61, k).astype(np.float32)
        v[::32] = 127.0
        return v
    x1, x2 = ints(), ints()
    x2[::32] = 0.0
    x12 = x1 + x2  # still has max 127 per block -> d == 1, exact
    y1 = w.matmul_vec(ca.HipTensor.new(x1, [k], hdev)).export().astype(np.float64)
    y2 = w.matmul_vec(ca.HipTensor.new(x2 + np.where(np.arange(k) % 32 == 0, 127.0, 0.0).astype(np.float32), [k], hdev)).export().astype(np.float64)
    y0 = w.matmul_vec(ca.HipTensor.new(np.where(np.arange(k) % 32 == 0, 127.0, 0.0).astype(np.float32), [k], hdev)).export().astype(np.float64)
    y12 = w.matmul_vec(ca.HipTensor.new(x12, [k], hdev)).export().astype(np.float64)
    scale = np.abs(y1) + np.abs(y2) + np.abs(y0) + np.abs(y12) + 1.0
    assert np.all(np.abs(y12 - (y1 + (y2 - y0))) <= 2e-4 * scale)
    rb = 18 * (k // 32)
    xq = o.quantize(x1, o.Q8_0)
    for row in (0, 7777, m - 1):
        ref = o.vec_dot(raw[row * rb:(row + 1) * rb], typ, xq, k)
        assert abs(y1[row] - ref) <= 1e-4 * (abs(ref) + 1.0)


# ---- strict order: every matmul_vec output equals the oracle's scalar loop bit for bit ---------------------------------------------
STRICT_FORMATS = FORMATS + ["F32", "F16"]


def _strict_cases(fmt):
    if fmt in ("F32", "F16"):
        return [(37, 64), (5, 2080)]
    # ragged rows (fewer than a wave's two, odd counts), k of 1 / 9 / 65 / 128 / 448 blocks, one super-block .. 56
    return [(3, 256), (5, 768), (129, 4096), (64, 14336)] if fmt in ("Q2_K", "Q3_K", "Q4_K", "Q5_K", "Q6_K", "Q8_K") else \
        [(1, 32), (3, 288), (7, 2080), (129, 4096), (64, 14336)]


def _strict_gemv_equals_oracle(ca, fmt):
    dev = ca.HipTensorDevice(0, False, 0, True)
    odev = o.OracleDevice(thread_num=2)
    for (m, k) in _strict_cases(fmt):
        typ, raw, x = make(fmt, m, k, 3 * m + k)
        w = ca.HipTensor.from_cpu(raw, [m, k], getattr(ca.GGMLType, HT[fmt]), dev)
        got = w.matmul_vec(ca.HipTensor.new(x, [k], dev)).export()
        ref = o.OracleTensor.from_bytes(raw, typ, [m, k], odev).matmul_vec(o.OracleTensor.new(x, [k], odev)).export()
        assert np.array_equal(got.view(np.uint32), ref.view(np.uint32)), f"{fmt} ({m},{k})"
        # a batch of rhs rows (the strict prompt pass): row by row the same
        xb = np.stack([x, -x, x * np.float32(0.5)]).astype(np.float32)
        gb = w.matmul_vec(ca.HipTensor.new(xb.reshape(-1), [3, k], dev)).export().reshape(3, m)
        rb = o.OracleTensor.from_bytes(raw, typ, [m, k], odev).matmul_vec(o.OracleTensor.new(xb.reshape(-1), [3, k], odev)).export().reshape(3, m)
        assert np.array_equal(gb.view(np.uint32), rb.view(np.uint32)), f"{fmt} ({m},{k}) batched"


@pytest.mark.parametrize("fmt", STRICT_FORMATS)
def test_strict_order_gemv_is_bit_exact(ca, fmt):
    """CRABML_HIP_FLAG_STRICT_ORDER: Q4_0 / Q8_0 / Q4_1 / Q5_0 / Q5_1 / Q2_K / Q8_K through the streaming kernels that park the block
    terms in LDS and add them in block order (k_gemv_exact_*), the rest through the one-thread-per-row kernel."""
    _strict_gemv_equals_oracle(ca, fmt)


def test_strict_order_scalar_kernel_still_covers_every_format():
    """the one-thread-per-row kernel (k_gemv_strict) behind the test hook: the two strict implementations agree with the oracle
    independently (a subprocess: the hook is read once per process)"""
    import os
    import subprocess
    import sys
    code = ("import sys; sys.path.insert(0, '.')\n"
            "import crabml_amd as ca\n"
            "from tests import test_hip_gemv as t\n"
            "for f in t.STRICT_FORMATS:\n"
            "    t._strict_gemv_equals_oracle(ca, f)\n"
            "print('scalar kernel ok')\n")
    env = dict(os.environ, CRABML_HIP_TEST_HOOKS="1", CRABML_HIP_STRICT_SCALAR="1")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    p = subprocess.run([sys.executable, "-c", code], cwd=root, env=env, capture_output=True, text=True, timeout=600)
    assert p.returncode == 0 and "scalar kernel ok" in p.stdout, (p.stdout[-1500:], p.stderr[-3000:])

"""GPU parity, end to end: Llama2Runner<HipTensor> (C++ host mirror -> C ABI -> HIP) against the oracle's
restatement of Llama2Runner<CpuTensor> on the same synthetic GGUF-layout weights.
Follows the reference's own CPU-vs-device test (crabml-llama2/src/llama2.rs:738-797): named debug tensors
first, then the greedy token stream.

Two gates:
  * STRICT-ORDER device (CRABML_HIP_FLAG_STRICT_ORDER): logits are BIT-IDENTICAL to the oracle (= the
    default, non-SIMD build of the reference) at every step, for every weight format.  Every op of the
    backend is bit-exact (test_hip_ops.py) and the strict GEMV adds block terms in the scalar-loop order.
  * FAST device (default kernels): the only fp divergence is the GEMV block-term summation order
    (<= 2e-5 * sum|w_i x_i| per output, test_hip_gemv.py).  End to end the reference's TRUNCATING
    activation quantizer amplifies such 1-ulp differences into +-1 quant flips, so its own scalar and
    AVX2 builds already differ by ~1e-2 of max|logit| (tests/test_oracle_runner.py).  Stated tolerance:
    max|hip - oracle| <= 3e-2 * max|oracle logit|, teacher-forced on the oracle's token stream."""
import numpy as np
import pytest

from crabml_amd import synth
from oracle import oracle as o
from tests.helpers import check_fast, to_oracle

pytestmark = pytest.mark.gpu

PROMPT = [1, 365, 400, 282]
LOGIT_TOL = 3e-2



def run_pair(ca, model, kv_f16, steps, debug=False, seq_len=64, strict=False):
    hdev = ca.HipTensorDevice(0, debug, 0, strict)
    odev = o.OracleDevice(thread_num=4, use_avx2=False, debug_named_tensors=debug)
    hconf, hw = synth.to_hip(model, hdev)
    oconf, ow = to_oracle(model, odev)
    hr = ca.Llama2Runner(hconf, hw, hdev, seq_len, kv_f16)
    orr = o.OracleLlamaRunner(oconf, ow, odev, seq_len, kv_f16)
    h_logits, o_logits = [], []
    pos = 0
    tok_h = tok_o = None
    toks = list(PROMPT)
    ids_h, ids_o = [], []
    for step in range(len(PROMPT) + steps - 1):
        t = toks[step] if step < len(PROMPT) else tok_o
        lh = hr.forward([t], pos).copy()
        lo = orr.forward([t], pos).copy()
        h_logits.append(lh)
        o_logits.append(lo)
        pos += 1
        if step >= len(PROMPT) - 1:
            tok_h, tok_o = o.argmax_last(lh), o.argmax_last(lo)
            ids_h.append(tok_h)
            ids_o.append(tok_o)
    return hdev, odev, h_logits, o_logits, ids_h, ids_o


def rel_errs(hl, ol):
    return np.array([np.max(np.abs(lh - lo)) / np.max(np.abs(lo)) for lh, lo in zip(hl, ol)])


@pytest.mark.parametrize("fmt", ["Q4_0", "Q8_0"])
@pytest.mark.parametrize("kv_f16", [False, True])
def test_15m_shape_decode_parity(ca, fmt, kv_f16):
    """Fast kernels, teacher-forced on the oracle's tokens.  Layer-0 / position-0 tensors (before the
    truncating re-quantization has amplified anything) are tight; later ones carry the stated tolerance."""
    model = synth.build_model(synth.SHAPES["15m"], synth.TYPE_BY_NAME[fmt], seed=20250103)
    hdev, odev, hl, ol, ids_h, ids_o = run_pair(ca, model, kv_f16, steps=12, debug=True)
    # llama2.rs:762-778 pattern (their eps: 1e-3 / 1e-7 / 1e-2); layer-0 inputs are bit-identical here
    a = hdev.dump_debug_tensor("attn_rmsnorm:0:0")
    b = odev.dump_debug_tensor("attn_rmsnorm:0:0")
    assert np.array_equal(a.view(np.uint32), b.view(np.uint32))
    assert np.allclose(hdev.dump_debug_tensor("x_debug:0:0"), odev.dump_debug_tensor("x_debug:0:0"), atol=1e-3, rtol=0)
    # (the reference's own CPU-vs-GPU test allows 1e-2 absolute on final_rmsnorm:0 for an F32 model,
    #  llama2.rs:780-784; with quantized weights every layer re-quantizes, hence the growth below)
    for name, tol in (("attn_out:0:0", 1e-3), ("ffn_out:0:0", 1e-2), ("final_rmsnorm:0", 1e-1),
                      ("ffn_out:5:0", 1e-1), ("final_rmsnorm:3", 1e-1)):
        x, y = hdev.dump_debug_tensor(name), odev.dump_debug_tensor(name)
        assert x is not None and y is not None, name
        assert np.max(np.abs(x - y)) <= tol * max(1.0, np.max(np.abs(y))), name
    err = rel_errs(hl, ol)
    check_fast(f"trait/15m/{fmt}/{'f16kv' if kv_f16 else 'f32kv'}", fmt, err)
    agree = sum(a == b for a, b in zip(ids_h, ids_o))
    assert ids_h[0] == ids_o[0] and agree >= len(ids_o) - 3, (ids_h, ids_o)


def test_fast_path_error_is_of_the_order_of_the_references_own_spread(ca):
    """Ties the fast path's end-to-end tolerance to the reference itself: on the same model and token
    stream, compare (hip fast vs scalar oracle) with (AVX2-order oracle vs scalar oracle)."""
    if not o.lib().co_have_avx2():
        pytest.skip("no avx2 on this host")
    model = synth.build_model(synth.SHAPES["15m"], synth.Q8_0, seed=20250103)
    _, _, hl, ol, _, ids_o = run_pair(ca, model, True, steps=8)
    odev2 = o.OracleDevice(thread_num=4, use_avx2=True)
    oconf, ow = to_oracle(model, odev2)
    r2 = o.OracleLlamaRunner(oconf, ow, odev2, 64, True)
    toks = list(PROMPT) + ids_o[:-1]
    al = [r2.forward([t], i).copy() for i, t in enumerate(toks)]
    e_hip, e_ref = rel_errs(hl, ol), rel_errs(al, ol)
    assert np.median(e_hip) <= 10 * np.median(e_ref) + 1e-3, (e_hip, e_ref)
    assert np.max(e_hip) <= 10 * np.max(e_ref) + 1e-3, (e_hip, e_ref)


@pytest.mark.parametrize("fmt", ["Q4_0", "Q8_0"])
@pytest.mark.parametrize("kv_f16", [False, True])
def test_15m_shape_strict_order_is_bit_exact(ca, fmt, kv_f16):
    model = synth.build_model(synth.SHAPES["15m"], synth.TYPE_BY_NAME[fmt], seed=20250103)
    hdev, odev, hl, ol, ids_h, ids_o = run_pair(ca, model, kv_f16, steps=8, debug=True, strict=True)
    for name in ("x_debug:0:0", "attn_out:0:0", "ffn_out:0:0", "ffn_out:5:2", "final_rmsnorm:0", "final_rmsnorm:6"):
        x, y = hdev.dump_debug_tensor(name), odev.dump_debug_tensor(name)
        assert np.array_equal(x.view(np.uint32), y.view(np.uint32)), name
    for step, (lh, lo) in enumerate(zip(hl, ol)):
        assert np.array_equal(lh.view(np.uint32), lo.view(np.uint32)), f"logits differ at step {step}"
    assert ids_h == ids_o


@pytest.mark.parametrize("fmt", ["Q4_0", "Q8_0", "Q4_1", "Q5_0", "Q5_1", "Q2_K", "Q3_K", "Q4_K", "Q5_K", "Q6_K", "Q8_K", "F32", "F16"])
@pytest.mark.parametrize("kv_f16", [False, True])
def test_gqa_shape_all_formats_strict_order_is_bit_exact(ca, fmt, kv_f16):
    model = synth.build_model(synth.SHAPES["tiny-gqa"], synth.TYPE_BY_NAME[fmt], seed=7)
    _, _, hl, ol, ids_h, ids_o = run_pair(ca, model, kv_f16, steps=6, strict=True)
    for step, (lh, lo) in enumerate(zip(hl, ol)):
        assert np.array_equal(lh.view(np.uint32), lo.view(np.uint32)), f"{fmt}: logits differ at step {step}"
    assert ids_h == ids_o


@pytest.mark.parametrize("fmt", ["Q4_0", "Q8_0", "Q4_1", "Q5_0", "Q5_1", "Q2_K", "Q3_K", "Q4_K", "Q5_K", "Q6_K", "Q8_K", "F32", "F16"])
def test_gqa_shape_all_formats(ca, fmt):
    """n_heads != n_kv_heads: exercises the GQA broadcast of batch_matmul with an f16 KV cache
    (bi / (ba/bb), batch_matmul.rs:89-91) for every weight format of the hot path."""
    model = synth.build_model(synth.SHAPES["tiny-gqa"], synth.TYPE_BY_NAME[fmt], seed=7)
    _, _, hl, ol, ids_h, ids_o = run_pair(ca, model, True, steps=6)
    err = rel_errs(hl, ol)
    check_fast(f"trait/tiny-gqa/{fmt}/f16kv", fmt, err)
    assert ids_h[0] == ids_o[0]


def test_gqa_f32_kv_uses_the_reference_modulo_broadcast(ca):
    """With an F32 cache the reference broadcasts kv heads as bi % n_kv (batch_matmul.rs:61-67), not
    bi / group -- reproduced, so outputs still agree with the oracle."""
    model = synth.build_model(synth.SHAPES["tiny-gqa"], synth.Q8_0, seed=9)
    _, _, hl, ol, ids_h, ids_o = run_pair(ca, model, False, steps=4, strict=True)
    for lh, lo in zip(hl, ol):
        assert np.array_equal(lh.view(np.uint32), lo.view(np.uint32))
    assert ids_h == ids_o


def test_cpp_generate_greedy_matches_stepwise(ca):
    model = synth.build_model(synth.SHAPES["tiny-gqa"], synth.Q4_0, seed=3)
    hdev = ca.HipTensorDevice(0, False)
    conf, w = synth.to_hip(model, hdev)
    r1 = ca.Llama2Runner(conf, w, hdev, 64, True)
    ids = r1.generate_greedy(PROMPT, 8)
    r2 = ca.Llama2Runner(conf, w, hdev, 64, True)
    pos = 0
    for t in PROMPT:
        lg = r2.forward([t], pos)
        pos += 1
    exp = []
    for _ in range(8):
        t = o.argmax_last(lg)
        exp.append(t)
        lg = r2.forward([t], pos)
        pos += 1
    assert list(ids) == exp
    assert r1.kv_cache_len() == len(PROMPT) + 7

"""The HEADLINE configuration against the oracle: the full Llama-3-8B shape (32 layers, dim 4096, hidden 14336, GQA 32/8,
vocab 128256), all-Q4_0 (BASELINE config C3) and all-Q4_K (C4), f16 KV cache -- the model bench.py times.

The 8B shape takes code paths the small test models never reach: two rows per wave from 8192 rows, split 32-row chunks in
the wo / ffn_down norm epilogue, 128 / 256 co-resident workgroups in the granule gather, the 128256-row classifier.  So
the comparisons the small shapes get are repeated here at full size, against `OracleLlamaRunner` (scalar order = the
reference's default build), teacher-forced on fixed tokens:

  * STRICT-order device: logits BIT-IDENTICAL to the oracle at every step (fused step in strict mode, and the per-op
    trait path for the first position);
  * FAST device (the benchmarked 5-kernel layers under the hipGraph): the observed error is asserted against a per-format
    tolerance that is a small multiple of what is measured (see FAST_TOL), and greedy tokens are compared;
  * the classifier GEMV (128256 x 4096), every row: strict == oracle, fast within the f32 re-association bound.
The oracle decodes this model at a few tokens/s on the host cores, which is what bounds the number of positions."""
import os

import numpy as np
import pytest

from crabml_amd import synth
from oracle import oracle as o
from tests.helpers import EXACT_NORM, GEMV_REL, to_oracle

pytestmark = pytest.mark.gpu

TOKENS = [1, 365, 400, 282, 9906]  # teacher-forced positions 0..4
SEQ = 32
# max over the steps of max|hip - oracle| / max|oracle logit| that the FAST path may show, per weight format: 4x the
# largest value observed on MI355X for this seed (4.9e-4 for Q4_0, 5.0e-4 for Q4_K; the test records what it sees in
# gpurun_out/headline_parity.json, last committed copy: tests/golden/headline_parity_observed.json)
FAST_TOL = {"Q4_0": 2e-3, "Q4_K": 2e-3}
# the same for the last-row logits of a 200-token prompt pass (test_fast_prompt_pass_against_the_oracle_at_the_8b_shape)
PROMPT_TOL = {"Q4_0": 2e-3, "Q4_K": 2e-3}  # observed (200 rows): f16 pass 4.6e-4 / 3.9e-4, int8 pass 5.9e-4 / 4.4e-4
_RESULTS = {}


def _threads():
    return max(16, min(64, os.cpu_count() or 16))


@pytest.fixture(scope="module", params=["Q4_0", "Q4_K"])
def headline(request, ca):
    fmt = request.param
    model = synth.build_model(synth.SHAPES["llama3-8b"], synth.TYPE_BY_NAME[fmt], seed=8)
    odev = o.OracleDevice(thread_num=_threads(), use_avx2=False)
    oconf, ow = to_oracle(model, odev)
    orr = o.OracleLlamaRunner(oconf, ow, odev, SEQ, True)
    ref = [orr.forward([t], i).copy() for i, t in enumerate(TOKENS)]
    del orr
    yield fmt, model, ref
    del model, ref


def _write_results():
    try:
        import json

        os.makedirs("gpurun_out", exist_ok=True)
        with open(os.path.join("gpurun_out", "headline_parity.json"), "w") as f:
            json.dump(_RESULTS, f, indent=1)
    except OSError:
        pass


def test_strict_device_is_bit_identical_to_the_oracle_at_the_8b_shape(ca, headline):
    fmt, model, ref = headline
    dev = ca.HipTensorDevice(0, False, 0, True)
    conf, w = synth.to_hip(model, dev)
    f = ca.HipLlamaRunner(conf, w, dev, SEQ, True)
    for i, t in enumerate(TOKENS):
        lg = f.forward(t, i)
        assert np.array_equal(lg.view(np.uint32), ref[i].view(np.uint32)), f"{fmt}: strict fused logits differ at position {i}"
    # the per-op trait path (Llama2Runner<HipTensor> unchanged), first two positions
    r = ca.Llama2Runner(conf, w, dev, SEQ, True)
    for i, t in enumerate(TOKENS[:2]):
        lg = r.forward([t], i)
        assert np.array_equal(lg.view(np.uint32), ref[i].view(np.uint32)), f"{fmt}: strict trait-path logits differ at position {i}"
    # and the batched prefill of the same tokens (strict: bit-identical to the token loop)
    p = ca.HipLlamaRunner(conf, w, dev, SEQ, True)
    lg = p.prefill(TOKENS)
    assert np.array_equal(lg.view(np.uint32), ref[-1].view(np.uint32)), f"{fmt}: strict prefill logits differ"


def test_fast_fused_path_against_the_oracle_at_the_8b_shape(ca, headline):
    fmt, model, ref = headline
    dev = ca.HipTensorDevice(0, False, 0, False)
    conf, w = synth.to_hip(model, dev)
    f = ca.HipLlamaRunner(conf, w, dev, SEQ, True)  # hipGraph, norm epilogue: exactly what bench.py times
    errs, med, agree, gaps = [], [], [], []
    for i, t in enumerate(TOKENS):
        lg = f.forward(t, i)
        d = np.abs(lg.astype(np.float64) - ref[i].astype(np.float64))
        scale = float(np.max(np.abs(ref[i])))
        errs.append(float(np.max(d)) / scale)
        med.append(float(np.median(d)) / scale)
        a_h, a_o = o.argmax_last(lg), o.argmax_last(ref[i])
        agree.append(a_h == a_o)
        # when the greedy token differs, the oracle's own top-2 gap must be inside the observed error (a tie, not a bug)
        gaps.append(float(ref[i][a_o] - ref[i][a_h]) / scale)
    _RESULTS[f"fast/{fmt}"] = {"max_rel_logit_err": errs, "median_rel_logit_err": med, "tokens_equal": agree,
                               "oracle_gap_where_different": gaps, "tolerance": FAST_TOL[fmt], "positions": len(TOKENS)}
    _write_results()
    assert max(errs) <= FAST_TOL[fmt], (fmt, errs)
    assert errs[0] <= 1e-3, (fmt, errs)  # position 0, before any re-quantization has amplified anything: tight
    for i, ok in enumerate(agree):
        assert ok or gaps[i] <= 2 * errs[i], (fmt, i, agree, gaps, errs)
    assert sum(agree) >= len(TOKENS) - 1, (fmt, agree)
    # the per-op fast path and the eager (graph-less) step give the same bits as the graph replay
    e = ca.HipLlamaRunner(conf, w, dev, SEQ, True, False)
    g = ca.HipLlamaRunner(conf, w, dev, SEQ, True)
    for i, t in enumerate(TOKENS[:2]):
        assert np.array_equal(e.forward(t, i).view(np.uint32), g.forward(t, i).view(np.uint32))


def test_classifier_gemv_all_rows_at_the_8b_shape(ca, headline):
    """output.weight (128256 x 4096) x one activation row: every output row compared."""
    fmt, model, _ = headline
    t = model.tensors["output.weight"]
    m, k = t.shape
    rng = np.random.default_rng(11)
    x = rng.standard_normal(k).astype(np.float32)
    odev = o.OracleDevice(thread_num=_threads(), use_avx2=False)
    ref = o.OracleTensor.from_bytes(t.data, t.typ, [m, k], odev).matmul_vec(o.OracleTensor.new(x, [k], odev)).export()
    tmap = {synth.Q4_0: ca.GGMLType.Q4_0, synth.Q4_K: ca.GGMLType.Q4K}
    sdev = ca.HipTensorDevice(0, False, 0, True)
    got_s = ca.HipTensor.from_cpu(t.data, [m, k], tmap[t.typ], sdev).matmul_vec(ca.HipTensor.new(x, [k], sdev)).export()
    assert np.array_equal(got_s.view(np.uint32), ref.view(np.uint32)), f"{fmt}: strict classifier GEMV differs from the oracle"
    fdev = ca.HipTensorDevice(0, False, 0, False)
    got_f = ca.HipTensor.from_cpu(t.data, [m, k], tmap[t.typ], fdev).matmul_vec(ca.HipTensor.new(x, [k], fdev)).export()
    # bound = GEMV_REL * sum_i |w_i||x_i| per row, from the dequantized weights, 8192 rows at a time
    rb = synth.BLOCK_BYTES[t.typ] * (k // synth.BLOCK_ELEMS[t.typ])
    ax = np.abs(x).astype(np.float32)
    worst = 0.0
    for r0 in range(0, m, 8192):
        r1 = min(m, r0 + 8192)
        wd = np.abs(o.dequantize(t.data[r0 * rb:r1 * rb], t.typ).reshape(r1 - r0, k))
        bound = (wd @ ax).astype(np.float64) * GEMV_REL * (8 if fmt == "Q4_K" else 1) + 1e-30
        err = np.abs(got_f[r0:r1].astype(np.float64) - ref[r0:r1].astype(np.float64))
        worst = max(worst, float(np.max(err / bound)))
    _RESULTS[f"classifier/{fmt}"] = {"rows": m, "max_err_over_bound": worst}
    _write_results()
    assert worst <= 1.0, (fmt, worst)


@pytest.mark.parametrize("fmt", ["Q4_0", "Q8_0"])
def test_zero_mean_weights_at_the_8b_shape_strict_exact_and_fast_inside_the_references_own_spread(ca, fmt):
    """The hard case for anything that re-associates or re-rounds: ZERO-MEAN weights (Q4_0 blocks with d of either sign; Q8_0's
    random int8 levels are zero-mean as drawn).  The benchmark's d > 0 blocks carry a common-mode component that makes relative
    logit errors look small (5e-4); here the reference's OWN two builds -- scalar and AVX2 order of the block dots -- are 7-10 % of
    max|logit| apart after 32 layers (one ulp moves a block's largest element between the levels 126 and 127 of the truncating
    quantizer, buf_q8_0.rs:119-124).  So: the strict-order device must still be BIT-IDENTICAL (fused entry point and the unchanged
    runner through the queue), and the fast step -- hop-free norm and all -- must sit inside k = 1.5 x the reference's own
    scalar-vs-AVX2 distance, computed here, on the same model and tokens (not a constant; round 5 allowed 2 x, observed: 1.28 x for
    Q4_0, 1.02 x for Q8_0 -- a change that moves it fails loudly).  Greedy-token agreement is recorded three ways -- the fast step
    and the reference's AVX2 build, each against the scalar oracle, and against each other -- in gpurun_out/headline_parity.json."""
    model = synth.build_model(synth.SHAPES["llama3-8b"], synth.TYPE_BY_NAME[fmt], seed=8)
    if fmt == "Q4_0":
        synth.flip_scale_signs(model, 5)
    toks = TOKENS[:4]
    outs = []
    for avx2 in (False, True):
        odev = o.OracleDevice(thread_num=_threads(), use_avx2=avx2)
        oconf, ow = to_oracle(model, odev)
        orr = o.OracleLlamaRunner(oconf, ow, odev, SEQ, True)
        outs.append([orr.forward([t], i).copy() for i, t in enumerate(toks)])
        del orr, ow
    ref, avx = outs
    rel = lambda a, b: float(np.max(np.abs(a.astype(np.float64) - b.astype(np.float64))) / np.max(np.abs(b)))  # noqa: E731
    spread = [rel(a, r) for a, r in zip(avx, ref)]
    sdev = ca.HipTensorDevice(0, False, 0, True)
    sconf, sw = synth.to_hip(model, sdev)
    s = ca.HipLlamaRunner(sconf, sw, sdev, SEQ, True)
    q = ca.Llama2Runner(sconf, sw, sdev, SEQ, True)  # the reference's unchanged runner: recorded calls -> fused strict segments
    for i, t in enumerate(toks):
        assert np.array_equal(s.forward(t, i).view(np.uint32), ref[i].view(np.uint32)), f"{fmt}: strict fused step, position {i}"
        assert np.array_equal(q.forward([t], i).view(np.uint32), ref[i].view(np.uint32)), f"{fmt}: strict runner through the queue, position {i}"
    assert sdev.lazy_stats()["fused_tokens"] == len(toks)
    del s, q, sw
    fdev = ca.HipTensorDevice(0)
    fconf, fw = synth.to_hip(model, fdev)
    res = {"reference_avx2_vs_scalar": spread}
    for name, flags in (("fast", 0), ("fast_exact_norm", EXACT_NORM)):
        f = ca.HipLlamaRunner(fconf, fw, fdev, SEQ, True, extra_flags=flags)
        lgs = [f.forward(t, i).copy() for i, t in enumerate(toks)]
        errs = [rel(lg, ref[i]) for i, lg in enumerate(lgs)]
        res[name] = errs
        am = lambda xs: [int(o.argmax_last(x)) for x in xs]  # noqa: E731
        t_fast, t_ref, t_avx = am(lgs), am(ref), am(avx)
        res[name + "_greedy_agreement"] = {"positions": len(toks), "fast_vs_scalar_oracle": sum(a == b for a, b in zip(t_fast, t_ref)),
                                           "reference_avx2_vs_scalar_oracle": sum(a == b for a, b in zip(t_avx, t_ref)),
                                           "fast_vs_reference_avx2": sum(a == b for a, b in zip(t_fast, t_avx))}
        del f
        assert max(errs) <= 1.5 * max(spread), (fmt, name, errs, spread)
    _RESULTS[f"zero_mean/{fmt}"] = res
    _write_results()


def test_fast_prompt_pass_against_the_oracle_at_the_8b_shape(ca, headline):
    """The fast prompt pass (crabml_hip_llama_prefill: the f16 weight GEMM with its epilogues, flash attention, the fused row kernels) at
    the headline shape: a 136-token prompt in one pass (two column tiles) -- every GEMM as the f16 kernel, gate | up with the SiLU * mul (+ quantizer)
    epilogue -- against the oracle's token loop over the same tokens (last-row logits), next to the same pass on the bit-exact int8
    GEMMs (CRABML_HIP_LLAMA_PREFILL_INT8_GEMM).  Bound: PROMPT_TOL of max |logit| (4 x observed on MI355X for this seed), and the f16
    pass no further from the oracle than 2 x the int8 pass + 1e-3; the greedy token agrees with the oracle's or sits inside the error."""
    fmt, model, _ = headline
    n = 136
    toks = [(31 * i + 7) % 50000 for i in range(n)]
    odev = o.OracleDevice(thread_num=_threads(), use_avx2=False)
    oconf, ow = to_oracle(model, odev)
    orr = o.OracleLlamaRunner(oconf, ow, odev, n + 8, True)
    ref = None
    for i, t in enumerate(toks):
        ref = orr.forward([t], i)
    ref = ref.copy()
    del orr
    dev = ca.HipTensorDevice(0, False, 0, False)
    conf, w = synth.to_hip(model, dev)
    scale = float(np.max(np.abs(ref)))
    out = {}
    for name, flags in (("f16", 0), ("int8", 524288)):
        r = ca.HipLlamaRunner(conf, w, dev, n + 8, True, extra_flags=flags)
        lg = np.array(r.prefill(toks))
        d = np.abs(lg.astype(np.float64) - ref.astype(np.float64))
        a_h, a_o = o.argmax_last(lg), o.argmax_last(ref)
        out[name] = {"max_rel_logit_err": float(np.max(d)) / scale, "median_rel_logit_err": float(np.median(d)) / scale,
                     "token_equal": bool(a_h == a_o), "oracle_gap_where_different": float(ref[a_o] - ref[a_h]) / scale}
        del r
    _RESULTS[f"fast_prompt_pass/{fmt}"] = dict(out, rows=n, tolerance=PROMPT_TOL[fmt])
    _write_results()
    ef, ei = out["f16"]["max_rel_logit_err"], out["int8"]["max_rel_logit_err"]
    assert ei <= PROMPT_TOL[fmt], (fmt, out)
    assert ef <= PROMPT_TOL[fmt] and ef <= 2 * ei + 1e-3, (fmt, out)
    for name in ("f16", "int8"):
        assert out[name]["token_equal"] or out[name]["oracle_gap_where_different"] <= 2 * out[name]["max_rel_logit_err"], (fmt, name, out)

"""GPU parity of the recorded-op queue (crabml_amd/csrc/lazy.hpp): the reference's UNCHANGED runner -- Llama2Runner<HipTensor>,
the C++ mirror of crabml-llama2/src/llama2.rs:184-281, 527-638, issuing one Tensor call after the other -- served by the fused
decode step, against (a) the same runner on a device that launches every call immediately (ABI version 1: the parity path of
rounds 1-4), (b) the fused step driven through its own entry point (HipLlamaRunner), (c) the oracle.

Strict-order device: all of them BIT-IDENTICAL, logits and KV-cache bytes, for every format and both cache types.  Fast device:
the queue's result is the fused step's (bit for bit) and within the format's stated tolerance of the oracle.
The host logic itself (which ops are replaced, aborts, deferred handles) is tested on the CPU: tests/test_lazy_queue.py."""
import numpy as np
import pytest

from crabml_amd import synth
from oracle import oracle as o
from tests.helpers import FAST_TOL, to_oracle
from tests.test_lazy_queue import F16, F32, caches, forward_py, forward_rs, ops_per_token

pytestmark = pytest.mark.gpu

TOKS = [1, 365, 400, 282, 7, 99, 512, 3]


def u32(a):
    return np.asarray(a, np.float32).view(np.uint32)


def kv_bytes(t):
    return np.asarray(t.export_raw())


@pytest.mark.parametrize("fmt", ["Q4_0", "Q8_0", "Q4_1", "Q4_K", "Q6_K", "F32"])
@pytest.mark.parametrize("kv_f16", [True, False])
def test_strict_queue_equals_per_op_launches_the_fused_entry_point_and_the_oracle(ca, fmt, kv_f16):
    model = synth.build_model(synth.SHAPES["tiny-gqa"], synth.TYPE_BY_NAME[fmt], seed=11)
    odev = o.OracleDevice(thread_num=4, use_avx2=False)
    oconf, ow = to_oracle(model, odev)
    orr = o.OracleLlamaRunner(oconf, ow, odev, 32, kv_f16)
    ref = [orr.forward([t], i).copy() for i, t in enumerate(TOKS)]

    lazy = ca.HipTensorDevice(0, False, 0, True)             # the default: recorded, served by the fused step
    eager = ca.HipTensorDevice(0, False, 0, True, "per-op")  # one launch per Tensor call
    conf, w = synth.to_hip(model, lazy)
    conf_e, w_e = synth.to_hip(model, eager)
    r = ca.Llama2Runner(conf, w, lazy, 32, kv_f16)
    re_ = ca.Llama2Runner(conf_e, w_e, eager, 32, kv_f16)
    f = ca.HipLlamaRunner(conf, w, lazy, 32, kv_f16)
    for i, t in enumerate(TOKS):
        a = r.forward([t], i).copy()
        b = re_.forward([t], i).copy()
        c = f.forward(t, i)
        assert np.array_equal(u32(a), u32(ref[i])), f"queue vs oracle, step {i}"
        assert np.array_equal(u32(b), u32(ref[i])), f"per-op launches vs oracle, step {i}"
        assert np.array_equal(u32(c), u32(ref[i])), f"fused entry point vs oracle, step {i}"
    st = lazy.lazy_stats()
    n = ops_per_token(model.shape.n_layers)
    assert st["fused_tokens"] == len(TOKS) and st["replayed"] == 0 and st["recorded"] == n * len(TOKS), st
    assert eager.lazy_stats()["recorded"] == 0


@pytest.mark.parametrize("fmt", ["Q4_0", "Q4_K"])
def test_kv_cache_bytes_and_kept_handles(ca, fmt):
    """The runner's OWN cache tensors are what the fused launches append to: their bytes equal the per-op launches'; the final
    normalized row, which forward() holds and never reads, is produced on demand with the per-op value."""
    model = synth.build_model(synth.SHAPES["tiny-gqa"], synth.TYPE_BY_NAME[fmt], seed=5)
    out = {}
    for mode in ("lazy", "per-op"):
        dev = ca.HipTensorDevice(0, False, 0, True, mode)
        conf, w = synth.to_hip(model, dev)
        kc, vc = caches(conf, dev, 32)
        res = []
        for i, t in enumerate(TOKS[:5]):
            lg, kept = forward_py(conf, w, dev, kc, vc, t, i, 32, eps=model.shape.rms_eps, keep="final")
            res.append((np.array(lg.export()), np.array(kept[0].export()), np.array(kept[1].export())))
        out[mode] = (res, [kv_bytes(k) for k in kc], [kv_bytes(v) for v in vc], dev.lazy_stats())
    lz, pe = out["lazy"], out["per-op"]
    for i in range(5):
        for j, what in enumerate(("logits", "x (final norm)", "x_final")):
            assert np.array_equal(u32(lz[0][i][j]), u32(pe[0][i][j])), f"{what}, step {i}"
    # only the 5 appended rows are defined (Tensor::alloc leaves f16 contents unspecified)
    s = model.shape
    hd = s.dim // s.n_heads
    for l in range(s.n_layers):
        for a, b in ((lz[1][l], pe[1][l]), (lz[2][l], pe[2][l])):
            a = a.view(np.uint16).reshape(s.n_kv_heads, 32, hd)[:, :5]
            b = b.view(np.uint16).reshape(s.n_kv_heads, 32, hd)[:, :5]
            assert np.array_equal(a, b), f"kv cache bytes, layer {l}"
    assert lz[3]["fused_tokens"] == 5 and lz[3]["deferred_bound"] == 15 and lz[3]["replayed"] == 0, lz[3]


@pytest.mark.parametrize("what", ["extra_op", "ffn_eps"])
def test_a_deviating_token_is_replayed_with_per_op_results(ca, what):
    """Segments of the deviating token had been enqueued (and had appended to the caches) when the stream left the template:
    the replay overwrites them, and the token's results are the per-op launches'."""
    model = synth.build_model(synth.SHAPES["tiny-gqa"], synth.Q4_0, seed=9)
    kw = {"extra_op_at": model.shape.n_layers - 1} if what == "extra_op" else {"ffn_eps": 1e-6}
    res = {}
    for mode in ("lazy", "per-op"):
        dev = ca.HipTensorDevice(0, False, 0, True, mode)
        conf, w = synth.to_hip(model, dev)
        kc, vc = caches(conf, dev, 32)
        lgs = []
        toks = (TOKS * 3)[:20]
        bad = (3, 5, 6, 9, 14, 15, 16)
        for i, t in enumerate(toks):
            extra = kw if i in bad else {}
            lg, _ = forward_py(conf, w, dev, kc, vc, t, i, 32, eps=model.shape.rms_eps, **extra)
            lgs.append(np.array(lg.export()))
        res[mode] = (lgs, dev.lazy_stats())
    # (a dropped shadow leaves granules of its in-launch hand-offs behind: the next token must not match them -- the step serial
    # their epochs derive from is set by the host at every token, lazy_ctx_begin)
    for i in range(20):
        assert np.array_equal(u32(res["lazy"][0][i]), u32(res["per-op"][0][i])), f"step {i}"
    st = res["lazy"][1]
    assert st["fused_tokens"] == 13 and st["aborts"] == 7, st


@pytest.mark.parametrize("fmt", ["Q4_0", "Q8_0", "Q4_K"])
def test_fast_queue_is_the_fused_step_and_within_tolerance(ca, fmt):
    model = synth.build_model(synth.SHAPES["tiny-gqa"], synth.TYPE_BY_NAME[fmt], seed=11)
    odev = o.OracleDevice(thread_num=4, use_avx2=False)
    oconf, ow = to_oracle(model, odev)
    orr = o.OracleLlamaRunner(oconf, ow, odev, 32, True)
    dev = ca.HipTensorDevice(0)
    conf, w = synth.to_hip(model, dev)
    r = ca.Llama2Runner(conf, w, dev, 32, True)
    f = ca.HipLlamaRunner(conf, w, dev, 32, True, False)  # eager launches, like the queue's segments
    errs = []
    for i, t in enumerate(TOKS):
        ref = orr.forward([t], i)
        a = r.forward([t], i).copy()
        c = f.forward(t, i)
        assert np.array_equal(u32(a), u32(c)), f"queue vs fused entry point, step {i}"
        errs.append(np.max(np.abs(a - ref)) / np.max(np.abs(ref)))
    assert np.median(errs) <= FAST_TOL[fmt][0] and max(errs) <= FAST_TOL[fmt][1], errs
    assert dev.lazy_stats()["fused_tokens"] == len(TOKS)


def test_15m_shape_generate_through_the_queue(ca):
    """BASELINE config C2's shape (head_dim 48, MHA): greedy generation of the unchanged runner, strict, against the oracle."""
    model = synth.build_model(synth.SHAPES["15m"], synth.Q4_0, seed=20250103)
    odev = o.OracleDevice(thread_num=4, use_avx2=False)
    oconf, ow = to_oracle(model, odev)
    orr = o.OracleLlamaRunner(oconf, ow, odev, 64, True)
    dev = ca.HipTensorDevice(0, False, 0, True)
    conf, w = synth.to_hip(model, dev)
    r = ca.Llama2Runner(conf, w, dev, 64, True)
    tok = 1
    for i in range(10):
        ref = orr.forward([tok], i)
        a = r.forward([tok], i)
        assert np.array_equal(u32(a), u32(ref)), f"step {i}"
        tok = o.argmax_last(ref)
    st = dev.lazy_stats()
    assert st["fused_tokens"] == 10 and st["replayed"] == 0, st


def test_single_ops_are_unaffected_by_the_queue(ca):
    """Outside a decode token the queue is the per-op launches in record order: in-place chains, views of one buffer, a buffer
    nothing has written (reads as zeros), and a handle dropped before the flush."""
    dev = ca.HipTensorDevice(0)
    x = np.arange(64, dtype=np.float32) / 7 - 3
    a = ca.HipTensor.new(x, [64], dev)
    b = a.dup().scale_inplace(2.0).add_inplace(a)
    tmp = b.dup().silu_inplace()  # dropped unread
    del tmp
    z = ca.HipTensor.alloc([8], F32, dev)
    assert np.array_equal(np.array(z.export()), np.zeros(8, np.float32))
    assert np.array_equal(np.array(b.export()), x * np.float32(2.0) + x)
    assert np.array_equal(np.array(a.export()), x)
    k = ca.HipTensor.alloc([2, 4, 8], F16, dev).resize(1, 0)
    row = ca.HipTensor.new(np.ones(16, np.float32), [2, 1, 8], dev)
    k.concatenate(row, 1)
    k.concatenate(row.dup().scale_inplace(3.0), 1)
    raw = np.asarray(k.export_raw()).view(np.float16).reshape(2, 4, 8)
    assert np.all(raw[:, 0] == 1) and np.all(raw[:, 1] == 3)
    st = dev.lazy_stats()
    assert st["replayed"] == st["recorded"] and st["fused_tokens"] == 0


def test_two_runners_taking_turns_on_one_device(ca):
    """Two Llama2Runner instances over the SAME weights, each with its own KV caches, advancing token by token in turns (two
    sequences served by one process): every switch is a token whose cache handles are not the learned context's -- it must be
    re-learned or replayed, never served by the other runner's context.  Strict device: both streams bit-identical to the per-op
    launches and to the oracle."""
    model = synth.build_model(synth.SHAPES["tiny-gqa"], synth.Q4_0, seed=23)
    odev = o.OracleDevice(thread_num=4, use_avx2=False)
    oconf, ow = to_oracle(model, odev)
    streams = ([1, 365, 400, 282, 7, 99], [3, 512, 99, 7, 282, 400])
    refs = []
    for toks in streams:
        orr = o.OracleLlamaRunner(oconf, ow, odev, 32, True)
        refs.append([orr.forward([t], i).copy() for i, t in enumerate(toks)])
    lazy = ca.HipTensorDevice(0, False, 0, True)
    conf, w = synth.to_hip(model, lazy)
    ra, rb = ca.Llama2Runner(conf, w, lazy, 32, True), ca.Llama2Runner(conf, w, lazy, 32, True)
    for i in range(6):
        a = ra.forward([streams[0][i]], i).copy()
        b = rb.forward([streams[1][i]], i).copy()
        assert np.array_equal(u32(a), u32(refs[0][i])), f"first runner, step {i}"
        assert np.array_equal(u32(b), u32(refs[1][i])), f"second runner, step {i}"
    # ... and a stretch of one runner alone afterwards is served by the fused step again
    st0 = lazy.lazy_stats()
    extra = [5, 6, 7, 8]
    orr = o.OracleLlamaRunner(oconf, ow, odev, 32, True)
    for i, t in enumerate(streams[0]):
        orr.forward([t], i)
    for j, t in enumerate(extra):
        assert np.array_equal(u32(ra.forward([t], 6 + j)), u32(orr.forward([t], 6 + j))), f"alone, step {j}"
    st1 = lazy.lazy_stats()
    assert st1["fused_tokens"] - st0["fused_tokens"] >= len(extra) - 1, (st0, st1)
    # each runner's context was built once and parked while the other was being served (round 6)
    assert st1["learned"] == 2 and st1["reactivated"] >= 10, st1


@pytest.mark.parametrize("fmt", ["Q4_0", "Q4_K"])
@pytest.mark.parametrize("kv_f16", [True, False])
def test_rust_lifetimes_same_fused_tokens_same_bits(ca, fmt, kv_f16):
    """The runner's call sequence with every handle released where rustc would release it (tests/test_lazy_queue.forward_rs: moves
    into consuming calls, statement temporaries, block scopes, caches through Option::take / replace -- llama2.rs:527-603): the
    matcher has only ever seen the C++ mirror's lifetimes.  Strict device: every token served by the fused step, logits bit-identical
    to the oracle and cache bytes to the per-op launches; a released runner's context is reaped and the next one learned afresh."""
    model = synth.build_model(synth.SHAPES["tiny-gqa"], synth.TYPE_BY_NAME[fmt], seed=29)
    odev = o.OracleDevice(thread_num=4, use_avx2=False)
    oconf, ow = to_oracle(model, odev)
    orr = o.OracleLlamaRunner(oconf, ow, odev, 32, kv_f16)
    ref = [orr.forward([t], i).copy() for i, t in enumerate(TOKS)]
    out = {}
    for mode in ("lazy", "per-op"):
        dev = ca.HipTensorDevice(0, False, 0, True) if mode == "lazy" else ca.HipTensorDevice(0, False, 0, True, "per-op")
        conf, w = synth.to_hip(model, dev)
        kc, vc = caches(conf, dev, 32, kv_f16)
        for i, t in enumerate(TOKS):
            lg = forward_rs(conf, w, dev, kc, vc, t, i, eps=model.shape.rms_eps)
            assert np.array_equal(u32(lg), u32(ref[i])), f"{mode}, step {i}"
        out[mode] = ([kv_bytes(k) for k in kc], [kv_bytes(v) for v in vc], dev.lazy_stats())
        if mode == "lazy":
            del kc, vc  # the "runner" goes: its context is the caches' only owner and is dropped at the next flush
            kc2, vc2 = caches(conf, dev, 32, kv_f16)
            for i, t in enumerate(TOKS[:3]):
                assert np.array_equal(u32(forward_rs(conf, w, dev, kc2, vc2, t, i, eps=model.shape.rms_eps)), u32(ref[i])), f"second cache set, step {i}"
            st2 = dev.lazy_stats()
            assert st2["reaped"] == 1 and st2["learned"] == 2 and st2["fused_tokens"] == len(TOKS) + 3, st2
    st = out["lazy"][2]
    assert st["fused_tokens"] == len(TOKS) and st["replayed"] == 0 and st["aborts"] == 0, st
    # only the appended rows are defined (Tensor::alloc leaves f16 contents unspecified)
    sh = model.shape
    hd, es = sh.dim // sh.n_heads, (np.uint16 if kv_f16 else np.uint32)
    for a, b in zip(out["lazy"][0] + out["lazy"][1], out["per-op"][0] + out["per-op"][1]):
        a = a.view(es).reshape(sh.n_kv_heads, 32, hd)[:, :len(TOKS)]
        b = b.view(es).reshape(sh.n_kv_heads, 32, hd)[:, :len(TOKS)]
        assert np.array_equal(a, b)

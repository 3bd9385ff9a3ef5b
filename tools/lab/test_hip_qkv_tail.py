"""The next layer's q/k/v GEMV as the tail of the ffn_down launch (CRABML_HIP_LLAMA_QKV_TAIL; crabml_amd/csrc/fused_ffn.hpp, QkvTail):
4 launches per layer instead of 5.  The tail uses k_qkv's lane -> block mapping, summation order and epilogue (rope.rs:47-63,
llama2.rs:244-256, 561-565, concatenate.rs:172-204), so the test is an equality test against the 5-launch step: every logit of
every step AND the KV-cache bytes, hipGraph replay and eager launches, f16 and f32 caches, at a small shape (split chunks forced)
and at the Llama-3-8B widths."""
import numpy as np
import pytest

from crabml_amd import synth

pytestmark = pytest.mark.gpu
QKV_TAIL = 2097152
SPLIT_ALWAYS = 16
TOKS = [1, 365, 400, 282, 7, 9, 11, 13, 21, 34]


def kv_equal(model, ra, rb, layer, which, kv16, n_pos, seq_len):
    """the FILLED region of the layer's K or V cache ([n_kv][seq_len][head_dim]; the rest is unwritten pool memory)"""
    s = model.shape
    es = 2 if kv16 else 4
    a, b = ra.debug_kv(layer, which, kv16), rb.debug_kv(layer, which, kv16)
    for h in range(s.n_kv_heads):
        lo = h * seq_len * s.head_dim * es
        n = n_pos * s.head_dim * es
        if not np.array_equal(a[lo:lo + n], b[lo:lo + n]):
            return False
    return True


def run(ca, model, flags, toks, kv16=True, use_graph=True, seq_len=64):
    dev = ca.HipTensorDevice(0)
    conf, w = synth.to_hip(model, dev)
    r = ca.HipLlamaRunner(conf, w, dev, seq_len, kv16, use_graph, True, extra_flags=flags)
    return [r.forward(t, i).copy() for i, t in enumerate(toks)], r


@pytest.mark.parametrize("kv16", [True, False])
@pytest.mark.parametrize("use_graph", [True, False])
def test_qkv_tail_equals_the_separate_launch_small_shape(ca, kv16, use_graph):
    model = synth.build_model(synth.SHAPES["tiny-gqa"], synth.Q4_0, seed=41)
    base, rb = run(ca, model, SPLIT_ALWAYS, TOKS, kv16, use_graph)
    got, rt = run(ca, model, SPLIT_ALWAYS | QKV_TAIL, TOKS, kv16, use_graph)
    for i, (a, b) in enumerate(zip(got, base)):
        assert np.array_equal(a.view(np.uint32), b.view(np.uint32)), f"step {i}"
    for layer in range(model.shape.n_layers):
        for which in (False, True):
            assert kv_equal(model, rt, rb, layer, which, kv16, len(TOKS), 64), (layer, which)


def test_qkv_tail_llama3_8b_widths(ca):
    model = synth.build_model(synth.SHAPES["llama3-8b"], synth.Q4_0, seed=42, n_layers=3)
    toks = [(37 * i + 11) % 128256 for i in range(24)]
    base, rb = run(ca, model, 0, toks, seq_len=64)
    got, rt = run(ca, model, QKV_TAIL, toks, seq_len=64)
    for i, (a, b) in enumerate(zip(got, base)):
        assert np.array_equal(a.view(np.uint32), b.view(np.uint32)), f"step {i}"
    for layer in range(3):
        for which in (False, True):
            assert kv_equal(model, rt, rb, layer, which, True, len(toks), 64), (layer, which)
    ids_b = rb.decode_greedy(5, 8)
    ids_t = rt.decode_greedy(5, 8)
    assert list(ids_b) == list(ids_t)

"""The recorded-op queue behind the Tensor entry points and the matcher that serves the reference's unchanged runner from the
fused decode step (crabml_amd/csrc/lazy.hpp) -- the HOST logic, on the CPU.

These tests run the real stack -- Llama2Runner<HipTensor> (the C++ mirror of crabml-llama2/src/llama2.rs:184-281, 527-638) ->
HipTensor -> the C ABI -> the queue / the matcher -- over the library's record-only test device (CRABML_HIP_FLAG_DRY, armed by
CRABML_HIP_TEST_HOOKS=1): calls are validated, recorded, matched and counted, nothing is computed.  What is asserted is which
ops were replaced by fused segments and which ran one launch at a time; what the segments COMPUTE is the GPU suite's business
(tests/test_hip_lazy.py: bit-identical logits against the per-op launches and the oracle)."""
import math
import os

import numpy as np
import pytest

os.environ["CRABML_HIP_TEST_HOOKS"] = "1"

import crabml_amd as ca  # noqa: E402
from crabml_amd import synth  # noqa: E402

F32, F16 = ca.GGMLType.F32, ca.GGMLType.F16


def dry(mode="dry", named=False):
    return ca.HipTensorDevice(0, named, 0, False, mode)


def tiny(dev, wtype=synth.Q4_0, shape="tiny-gqa"):
    model = synth.build_model(synth.SHAPES[shape], wtype, seed=1)
    conf, w = synth.to_hip(model, dev)
    return model.shape, conf, w


def ops_per_token(n_layers):
    return 1 + 26 * n_layers + 4  # embedding; 26 Tensor calls per layer; final norm (2), last-row copy, classifier


def test_the_unchanged_runner_is_served_by_the_fused_step():
    dev = dry()
    s, conf, w = tiny(dev)
    r = ca.Llama2Runner(conf, w, dev, 32, True)
    n = ops_per_token(s.n_layers)
    for i, t in enumerate([1, 2, 3, 4, 5]):
        r.forward([t], i)
        st = dev.lazy_stats()
        # every token -- the one the decode context is learned from included -- runs as 2 L + 1 fused segments
        assert st["recorded"] == n * (i + 1)
        assert st["fused_tokens"] == i + 1 and st["fused_ops"] == n * (i + 1)
        assert st["segments"] == (2 * s.n_layers + 1) * (i + 1)
        assert st["replayed"] == 0 and st["aborts"] == 0 and st["learned"] == 1
        assert st["deferred_bound"] == 0  # x / x_final of forward() were dropped unread: never bound, never computed
    assert r.kv_cache_len() == 5


@pytest.mark.parametrize("mode", ["dry-per-op"])
def test_no_fusion_mode_replays_every_op(mode):
    dev = dry(mode)
    s, conf, w = tiny(dev)
    r = ca.Llama2Runner(conf, w, dev, 32, True)
    for i, t in enumerate([1, 2, 3]):
        r.forward([t], i)
    st = dev.lazy_stats()
    assert st["replayed"] == st["recorded"] == 3 * ops_per_token(s.n_layers)
    assert st["fused_tokens"] == 0 and st["learned"] == 0


def test_f32_cache_and_other_formats_are_matched_too():
    for wtype in (synth.Q8_0, synth.Q4_1, synth.Q4_K, synth.F32):
        dev = dry()
        s, conf, w = tiny(dev, wtype)
        r = ca.Llama2Runner(conf, w, dev, 16, False)  # f32 KV cache
        for i, t in enumerate([3, 1, 2]):
            r.forward([t], i)
        st = dev.lazy_stats()
        assert st["fused_tokens"] == 3 and st["replayed"] == 0, (wtype, st)


def test_a_second_runner_on_the_same_weights_gets_its_own_context():
    dev = dry()
    s, conf, w = tiny(dev)
    n = ops_per_token(s.n_layers)
    r1 = ca.Llama2Runner(conf, w, dev, 32, True)
    for i in range(3):
        r1.forward([i + 1], i)
    r2 = ca.Llama2Runner(conf, w, dev, 32, True)  # other KV-cache buffers: the template's cache handles do not match
    for i in range(3):
        r2.forward([i + 1], i)
    st = dev.lazy_stats()
    assert st["learned"] == 2
    # r2's first token starts as a shadow of r1's template, deviates at the first concatenate (before anything was launched),
    # and is then the token the new context is learned from -- and served by
    assert st["aborts"] == 1 and st["fused_tokens"] == 6 and st["replayed"] == 0
    assert st["recorded"] == 6 * n
    # ... and going back to r1 takes its parked context back into service (round 6; before, every switch built a new one)
    r1.forward([9], 3)
    st = dev.lazy_stats()
    assert st["learned"] == 2 and st["reactivated"] == 1 and st["fused_tokens"] == 7 and st["replayed"] == 0


def test_named_tensor_snapshots_force_the_per_op_path():
    dev = dry("dry", named=True)  # with_name() exports: a flush in the middle of every layer
    s, conf, w = tiny(dev)
    r = ca.Llama2Runner(conf, w, dev, 32, True)
    for i, t in enumerate([1, 2]):
        r.forward([t], i)
    st = dev.lazy_stats()
    assert st["fused_tokens"] == 0 and st["replayed"] == st["recorded"] == 2 * ops_per_token(s.n_layers)
    assert dev.dump_debug_tensor("attn_out:0:1") is not None


# ---- the same op sequence issued from python, so that single calls can be perturbed -------------------------------
def forward_py(conf, w, dev, kc, vc, tok, pos, seq, eps=1e-5, ffn_eps=1e-5, extra_op_at=None, keep=None, scale=None):
    """Llama2Runner::forward (llama2.rs:184-281, 527-638), n_batch = 1, call for call; returns (logits tensor, kept handles)"""
    kept = []
    dim, hd = conf.embedding_dim, conf.head_size()
    nh, nkv = conf.n_heads, conf.n_kv_heads
    x = ca.HipTensor.alloc([1, dim], F32, dev)
    x.copy_rows_from(w.token_embed, [tok])
    for l in range(conf.n_layers):
        xo = x.dup()
        x = x.rms_norm_inplace(eps)
        x = x.mul_inplace(w.rms_att_weight[l])
        q = w.wq[l].matmul_vec(x)
        k = w.wk[l].matmul_vec(x)
        v = w.wv[l].matmul_vec(x)
        q = q.reshape([1, nh, hd])
        k = k.reshape([1, nkv, hd])
        q = q.rope_inplace(ca.RopeMode.Llama, pos, hd)
        k = k.rope_inplace(ca.RopeMode.Llama, pos, hd)
        if extra_op_at == l:
            k = k.scale_inplace(1.0)  # a call forward_llama does not make
        kc[l].concatenate(k.reshape([1, nkv, hd]).transpose([1, 0, 2]), 1)
        vc[l].concatenate(v.reshape([1, nkv, hd]).transpose([1, 0, 2]), 1)
        qc = q.reshape([1, nh, hd]).transpose([1, 0, 2]).contiguous().scale_inplace(scale if scale else 1.0 / math.sqrt(np.float32(hd)))
        korig = kc[l].strider()
        kt = kc[l].transpose([0, 2, 1])
        att = qc.batch_matmul(kt).softmax_inplace(2)
        kc[l] = kt.with_strider(korig)
        xa = att.batch_matmul(vc[l]).reshape([1, dim])
        if keep == ("xa", l):
            kept.append(xa)  # an intermediate the fused step never materializes escapes
        x = w.wo[l].matmul_vec(xa)
        x = x.add_inplace(xo)
        xo2 = x.dup()
        x = x.rms_norm_inplace(ffn_eps)
        x = x.mul_inplace(w.rms_ffn_weight[l])
        h1 = w.ffn_gate_weight[l].matmul_vec(x)
        h2 = w.ffn_up_weight[l].matmul_vec(x)
        h1 = h1.silu_inplace().mul_inplace(h2)
        x = w.ffn_down_weight[l].matmul_vec(h1)
        x = x.add_inplace(xo2)
        del xo, q, k, v, qc, kt, att, xa, xo2, h1, h2  # the locals of the Rust loop body go out of scope here
    x = x.rms_norm_inplace(eps)
    x = x.mul_inplace(w.rms_final_weight)
    xf = ca.HipTensor.alloc([dim], F32, dev)
    xf.copy_rows_from(x, [0])
    ow = w.output_weight if w.output_weight is not None else w.token_embed
    logits = ow.matmul_vec(xf)
    if keep == "final":
        kept += [x, xf]
    return logits, kept


def forward_rs(conf, w, dev, kc, vc, tok, pos, eps=1e-5):
    """The same call sequence with every handle released WHERE RUST WOULD release it (the C++ mirror and forward_py keep a loop body's
    locals until its end): values are moved into the consuming call (`q.reshape(..)` leaves no `q` behind), temporaries of a method
    chain die at the end of their statement, a block's locals at its closing brace, and the caches travel through
    `Option::take()` / `replace()` (llama2.rs:571-597: `self.key_cache[l]` is None while the scores are computed).  The recorded ops
    hold their own references, so none of this may change what the matcher sees -- same fused tokens, same bits."""
    dim, hd = conf.embedding_dim, conf.head_size()
    nh, nkv = conf.n_heads, conf.n_kv_heads
    x = ca.HipTensor.alloc([1, dim], F32, dev)
    x.copy_rows_from(w.token_embed, [tok])
    for l in range(conf.n_layers):
        x_attn_orig = x.dup()                                  # llama2.rs:229
        x = x.rms_norm_inplace(eps)                            # (in-place methods consume self and return it)
        x = x.mul_inplace(w.rms_att_weight[l])
        q = w.wq[l].matmul_vec(x)                              # :244-246
        k = w.wk[l].matmul_vec(x)
        v = w.wv[l].matmul_vec(x)
        q = q.reshape([1, nh, hd]).rope_inplace(ca.RopeMode.Llama, pos, hd)   # :251-256 (reshape consumes q: no handle stays behind)
        k = k.reshape([1, nkv, hd]).rope_inplace(ca.RopeMode.Llama, pos, hd)
        # ---- forward_multi_query_attention(q, k, v, ..) -- q, k, v MOVED in ----
        kv_k = k.reshape([1, nkv, hd]).transpose([1, 0, 2])    # :542-547 (`k.reshape` takes &self here: the parameter stays alive)
        kv_v = v.reshape([1, nkv, hd]).transpose([1, 0, 2])
        kc[l].concatenate(kv_k, 1)
        vc[l].concatenate(kv_v, 1)
        del kv_k, kv_v                                         # the block's closing brace (:555)
        q = q.reshape([1, nh, hd]).transpose([1, 0, 2]).contiguous().scale_inplace(1.0 / math.sqrt(np.float32(hd)))  # :561-565
        k_cache, kc[l] = kc[l], None                           # self.key_cache[l].take()  (:571)
        k_orig = k_cache.strider()
        k_cache = k_cache.transpose([0, 2, 1])                 # consumes the taken cache
        attn = q.batch_matmul(k_cache)
        attn = attn.softmax_inplace(2)
        kc[l] = k_cache.with_strider(k_orig)                   # replace(..)  (:578)
        del k_cache
        v_cache, vc[l] = vc[l], None                           # (:584)
        v_orig = v_cache.strider()
        x_with_attn = attn.batch_matmul(v_cache)
        x_with_attn = x_with_attn.reshape([1, dim])
        vc[l] = v_cache.with_strider(v_orig)                   # (:597)
        del v_cache
        x = w.wo[l].matmul_vec(x_with_attn)                    # the old x (the normalized row) is dropped by the assignment
        del q, attn, x_with_attn                               # the `let x = { .. }` block ends (:601)
        del k, v                                               # the function returns: its by-value parameters die
        x = x.add_inplace(x_attn_orig)                         # :266
        # ---- forward_ffn(x, ..): x MOVED in ----
        x_orig_ffn = x.dup()
        x = x.rms_norm_inplace(1e-5)
        x = x.mul_inplace(w.rms_ffn_weight[l])
        h1 = w.ffn_gate_weight[l].matmul_vec(x)
        h2 = w.ffn_up_weight[l].matmul_vec(x)
        h1 = h1.silu_inplace()
        h1 = h1.mul_inplace(h2)
        x = w.ffn_down_weight[l].matmul_vec(h1)                # (drops the normalized row)
        x = x.add_inplace(x_orig_ffn)
        del x_orig_ffn, h1, h2                                 # forward_ffn returns
        del x_attn_orig                                        # the loop body's scope ends
    x = x.rms_norm_inplace(eps)
    x = x.mul_inplace(w.rms_final_weight)
    x_final = ca.HipTensor.alloc([dim], F32, dev)
    x_final.copy_rows_from(x, [0])
    ow = w.output_weight if w.output_weight is not None else w.token_embed
    logits = ow.matmul_vec(x_final)
    out = np.array(logits.export())                            # forward() ends: x, x_final, logits die after the export
    del x, x_final, logits
    return out


def caches(conf, dev, seq, f16=True):
    mk = lambda: ca.HipTensor.alloc([conf.n_kv_heads, seq, conf.head_size()], F16 if f16 else F32, dev).resize(1, 0)  # noqa: E731
    return [mk() for _ in range(conf.n_layers)], [mk() for _ in range(conf.n_layers)]


def test_the_python_restatement_issues_the_runners_sequence():
    dev = dry()
    s, conf, w = tiny(dev)
    kc, vc = caches(conf, dev, 32)
    for i in range(3):
        lg, _ = forward_py(conf, w, dev, kc, vc, i + 1, i, 32, eps=s.rms_eps)
        lg.export()
    st = dev.lazy_stats()
    assert st["fused_tokens"] == 3 and st["replayed"] == 0 and st["recorded"] == 3 * ops_per_token(s.n_layers)


@pytest.mark.parametrize("what", ["extra_op", "ffn_eps", "scale", "pos"])
def test_a_deviating_token_runs_op_by_op(what):
    dev = dry()
    s, conf, w = tiny(dev)
    n = ops_per_token(s.n_layers)
    kc, vc = caches(conf, dev, 32)
    for i in range(2):
        forward_py(conf, w, dev, kc, vc, i + 1, i, 32, eps=s.rms_eps)[0].export()
    base = dev.lazy_stats()
    assert base["fused_tokens"] == 2
    kw = {}
    extra = 0
    if what == "extra_op":
        kw["extra_op_at"] = s.n_layers - 1  # the earlier layers' segments are already enqueued when the stream deviates
        extra = 1
    elif what == "ffn_eps":
        kw["ffn_eps"] = 1e-6  # the fused step hard-codes the literal 1e-5 of llama2.rs:611
    elif what == "scale":
        kw["scale"] = 0.25
    pos = 2
    if what == "pos":  # rope position != cache length
        x = ca.HipTensor.alloc([1, conf.embedding_dim], F32, dev)
        x.copy_rows_from(w.token_embed, [1])
        q = w.wq[0].matmul_vec(x.dup().rms_norm_inplace(s.rms_eps))
        q.reshape([1, conf.n_heads, conf.head_size()]).rope_inplace(ca.RopeMode.Llama, 5, conf.head_size())
        q.export()
        st = dev.lazy_stats()
        assert st["fused_tokens"] == 2 and st["replayed"] == 5 and st["aborts"] == 1
        return
    forward_py(conf, w, dev, kc, vc, 3, pos, 32, eps=s.rms_eps, **kw)[0].export()
    st = dev.lazy_stats()
    assert st["fused_tokens"] == 2 and st["aborts"] == 1
    assert st["replayed"] == n + extra  # the whole token, the ops whose segments had been enqueued included
    # the next regular token is served by the fused step again
    forward_py(conf, w, dev, kc, vc, 4, pos + 1, 32, eps=s.rms_eps)[0].export()
    st = dev.lazy_stats()
    assert st["fused_tokens"] == 3 and st["replayed"] == n + extra


def test_an_escaping_intermediate_takes_the_model_off_the_fused_step():
    dev = dry()
    s, conf, w = tiny(dev)
    n = ops_per_token(s.n_layers)
    kc, vc = caches(conf, dev, 32)
    forward_py(conf, w, dev, kc, vc, 1, 0, 32, eps=s.rms_eps)[0].export()
    lg, kept = forward_py(conf, w, dev, kc, vc, 2, 1, 32, eps=s.rms_eps, keep=("xa", 0))
    lg.export()
    st = dev.lazy_stats()
    # the host holds a handle whose value only the per-op launches produce: the token is replayed, nothing is committed
    assert st["fused_tokens"] == 1 and st["aborts"] == 1 and st["replayed"] == n
    kept[0].export()  # and is readable


def test_the_final_row_is_bound_only_when_somebody_keeps_it():
    dev = dry()
    s, conf, w = tiny(dev)
    kc, vc = caches(conf, dev, 32)
    forward_py(conf, w, dev, kc, vc, 1, 0, 32, eps=s.rms_eps)[0].export()
    assert dev.lazy_stats()["deferred_bound"] == 0
    lg, kept = forward_py(conf, w, dev, kc, vc, 2, 1, 32, eps=s.rms_eps, keep="final")
    lg.export()
    assert dev.lazy_stats()["deferred_bound"] == 0  # still only promised
    kept[0].export()  # x: read -> every promised handle the host still holds is bound: x, x_final (the context's residual stream) and
    st = dev.lazy_stats()  # the logits handle (the context's logits buffer; its export above was served from the pinned host copy)
    assert st["deferred_bound"] == 3 and st["fused_tokens"] == 2 and st["replayed"] == 0 and st["pinned_exports"] == 2
    # kept, never read, and the next token starts: bound before the residual stream moves on
    lg, kept2 = forward_py(conf, w, dev, kc, vc, 3, 2, 32, eps=s.rms_eps, keep="final")
    lg.export()
    forward_py(conf, w, dev, kc, vc, 4, 3, 32, eps=s.rms_eps)[0].export()
    st = dev.lazy_stats()
    assert st["deferred_bound"] == 6 and st["fused_tokens"] == 4
    del kept, kept2


def test_ops_before_a_token_run_first_and_ops_outside_tokens_are_replayed():
    dev = dry()
    s, conf, w = tiny(dev)
    kc, vc = caches(conf, dev, 32)
    forward_py(conf, w, dev, kc, vc, 1, 0, 32, eps=s.rms_eps)[0].export()
    a = ca.HipTensor.new(np.ones(64, np.float32), [64], dev)
    b = a.dup().scale_inplace(2.0).add_inplace(a)  # three recorded ops that belong to no token
    lg, _ = forward_py(conf, w, dev, kc, vc, 2, 1, 32, eps=s.rms_eps)
    st = dev.lazy_stats()
    assert st["replayed"] == 3 and st["fused_tokens"] == 2  # flushed when the token started; the token itself already committed
    lg.export()
    b.export()


def test_unsupported_models_stay_on_the_per_op_path():
    dev = dry()
    sh = synth.SHAPES["tiny-gqa"]
    model = synth.build_model(sh, synth.Q4_0, seed=1)
    conf, w = synth.to_hip(model, dev)
    kc, vc = caches(conf, dev, 32)
    # a neox-style rope is not the sequence of forward_llama
    x = ca.HipTensor.alloc([1, conf.embedding_dim], F32, dev)
    x.copy_rows_from(w.token_embed, [1])
    x.export()
    assert dev.lazy_stats()["learned"] == 0


def test_the_runners_greedy_sampler_keeps_the_last_maximum():
    """sampler.rs:109-116 (max_by keeps the last maximum); the mirror's two-pass form against the defining loop"""
    def ref(p):
        best = 0
        for i in range(1, len(p)):
            if not (p[best] > p[i]):
                best = i
        return best

    rng = np.random.default_rng(3)
    for n in (1, 7, 63, 64, 65, 100, 1000, 4099):
        for case in range(6):
            p = rng.standard_normal(n).astype(np.float32)
            if case == 1:
                p[rng.integers(0, n, size=3)] = p.max()  # ties
            elif case == 2:
                p[:] = 0.0
                p[rng.integers(0, n)] = -0.0  # +-0 compare equal
            elif case == 3:
                p[rng.integers(0, n)] = np.nan
            elif case == 4:
                p[rng.integers(0, n, size=2)] = np.inf
            elif case == 5:
                p = np.round(p * 2).astype(np.float32)  # many ties
            assert ca.sample_argmax(p) == ref(p), (n, case)


def test_rust_lifetimes_do_not_change_what_the_matcher_sees():
    """The op stream with every handle released where rustc would release it (forward_rs): every token is still served by the fused
    step -- the queue's own references keep operands alive, caches moved out of and back into their Option are the same handles."""
    dev = dry()
    s, conf, w = tiny(dev)
    n = ops_per_token(s.n_layers)
    kc, vc = caches(conf, dev, 32)
    for i in range(5):
        forward_rs(conf, w, dev, kc, vc, i + 1, i, eps=s.rms_eps)
        st = dev.lazy_stats()
        assert st["fused_tokens"] == i + 1 and st["replayed"] == 0 and st["aborts"] == 0, st
        assert st["recorded"] == n * (i + 1) and st["deferred_bound"] == 0
    # ... and it is the SAME stream as the C++ mirror's: a runner's context (other caches) is learned once, not per token
    assert dev.lazy_stats()["learned"] == 1


def test_runners_taking_turns_keep_their_contexts():
    """Two runners on one device advancing in turns: each runner's decode context is built once and parked while the other is served
    (round-5 review: exactly one context was kept and every switch rebuilt it -- allocations and a graph capture inside the flush)."""
    dev = dry()
    s, conf, w = tiny(dev)
    r1, r2 = ca.Llama2Runner(conf, w, dev, 32, True), ca.Llama2Runner(conf, w, dev, 32, True)
    for i in range(4):
        r1.forward([i + 1], i)
        r2.forward([i + 2], i)
    st = dev.lazy_stats()
    assert st["learned"] == 2, st                       # one context per runner, ever
    assert st["reactivated"] == 6, st                   # every later switch takes the parked one back
    assert st["fused_tokens"] == 8 and st["replayed"] == 0, st
    # a third and a fourth runner push the oldest context out (at most two wait beside the one being served) ...
    r3, r4 = ca.Llama2Runner(conf, w, dev, 32, True), ca.Llama2Runner(conf, w, dev, 32, True)
    r3.forward([1], 0)
    r4.forward([1], 0)
    assert dev.lazy_stats()["learned"] == 4
    r1.forward([9], 4)  # ... r1's was the oldest: built again
    st = dev.lazy_stats()
    assert st["learned"] == 5 and st["replayed"] == 0, st


def test_a_released_runner_or_model_releases_its_decode_context():
    """The decode context retains the weights and the runner's caches.  A host that drops its runner (or the whole model) gets that
    device memory back at the next flush: a context that is the only remaining owner of its caches or weights is destroyed."""
    dev = dry()
    s, conf, w = tiny(dev)
    r = ca.Llama2Runner(conf, w, dev, 32, True)
    for i in range(3):
        r.forward([i + 1], i)
    assert dev.lazy_stats()["reaped"] == 0
    del r                                              # the runner and its caches go; the weights stay
    a = ca.HipTensor.new(np.ones(8, np.float32), [8], dev)
    a.dup().scale_inplace(2.0).export()                # any flush
    st = dev.lazy_stats()
    assert st["reaped"] == 1, st  # (its private buffers and the cache blocks are back in the device's pool)
    # the weights are still usable, and a new runner is learned and served as before
    r = ca.Llama2Runner(conf, w, dev, 32, True)
    r.forward([1], 0)
    r.forward([2], 1)
    st = dev.lazy_stats()
    assert st["learned"] == 2 and st["fused_tokens"] == 5 and st["replayed"] == 2, st  # (replayed: the two ops outside any token)

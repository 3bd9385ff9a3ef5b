"""k_attn_flash -- the fast decode step's attention from `attn_long_from` cached positions on (split-KV, f32 exp and
accumulation, last-arriver merge; crabml_amd/csrc/fused_attention.hpp) -- pinned against FLOAT64 arithmetic on the same inputs.

The reference computes softmax(q K^T) V with q rounded to f16 (batch_matmul.rs:39), the f16 exp table of f16(s - max)
(softmax.rs:43-53, buf_f32.rs:29-35), probabilities rounded to f16 and a serial f16 accumulator over the positions
(buf_f16.rs:152-163).  The strict-order device and CRABML_HIP_LLAMA_EXACT_ATTENTION reproduce all of that bit for bit; the fast
kernel keeps only the first rounding (q -> f16; K and V are f16 in the cache) and does the rest in f32.  Stated tolerance of the
kernel against exact arithmetic on those f16 inputs: 2e-5 of max|out| (observed ~2e-6) -- it is the reference's f16 noise, not
the kernel's, that separates the two paths end to end (tests/test_hip_long_context_oracle.py, FLASH_TOL)."""
import numpy as np
import pytest

from crabml_amd import synth

pytestmark = pytest.mark.gpu


def f64_attention(q, k, v, n_heads, n_kv, hd, seq):
    q16 = q.astype(np.float16).astype(np.float64).reshape(n_heads, hd)
    kf = k.view(np.float16).astype(np.float64).reshape(n_kv, seq, hd)
    vf = v.view(np.float16).astype(np.float64).reshape(n_kv, seq, hd)
    grp = n_heads // n_kv
    out = np.zeros((n_heads, hd))
    for h in range(n_heads):
        s = kf[h // grp] @ q16[h]
        p = np.exp(s - s.max())
        out[h] = (p / p.sum()) @ vf[h // grp]
    return out.reshape(-1)


def run_case(ca, dev, rng, n_heads, n_kv, hd, seq, slices, spread=1.0, kind="normal"):
    q = (rng.standard_normal(n_heads * hd) * spread / np.sqrt(hd)).astype(np.float32)
    k = rng.standard_normal((n_kv, seq, hd)).astype(np.float16)
    v = rng.standard_normal((n_kv, seq, hd)).astype(np.float16)
    if kind == "spike":  # one position dominates, far from the slice that holds the running maximum first
        k[:, seq // 2] *= 6.0
    elif kind == "ties":  # every score equal: uniform probabilities over all slices
        k[:] = k[:, :1]
    elif kind == "dead":  # scores so low in all but the last slice that their weights underflow to 0 in the merge
        k[:, : seq - 1] = (-np.abs(k[:, : seq - 1].astype(np.float32)) * 3).astype(np.float16)
        q = np.abs(q) * 8
    got, got_ticket = dev.debug_flash_attention(q, k.view(np.uint16).reshape(-1), v.view(np.uint16).reshape(-1), n_heads, n_kv, hd, seq, slices)
    # the single-launch form (last arriver merges behind a ticket) performs the same operations in the same order
    assert np.array_equal(got.view(np.uint32), got_ticket.view(np.uint32)), (n_heads, n_kv, hd, seq, slices)
    ref = f64_attention(q, k.view(np.uint16), v.view(np.uint16), n_heads, n_kv, hd, seq)
    assert np.all(np.isfinite(got))
    return float(np.max(np.abs(got - ref)) / np.max(np.abs(ref)))


@pytest.mark.parametrize("n_heads,n_kv,hd", [(32, 8, 128), (8, 8, 128), (8, 4, 128), (8, 1, 128), (8, 2, 64), (8, 8, 64), (8, 1, 64)])
def test_flash_attention_equals_float64_arithmetic(ca, n_heads, n_kv, hd):
    dev = ca.HipTensorDevice(0)
    rng = np.random.default_rng(100 * n_kv + hd)
    worst = 0.0
    # 1 .. a few rows (one slice, ragged row groups), the slice thresholds (128 rows each), many slices, more rows than
    # 32 slices x one round of loads
    for seq in (1, 2, 3, 5, 8, 31, 127, 128, 129, 255, 256, 257, 1000, 1024, 2049, 4096, 9001):
        for slices in (1, 7, 32):
            worst = max(worst, run_case(ca, dev, rng, n_heads, n_kv, hd, seq, slices))
    assert worst <= 2e-5, worst
    print(f"flash vs float64, {n_heads} heads / {n_kv} kv x {hd}: worst {worst:.2e} of max|out|")


@pytest.mark.parametrize("kind", ["spike", "ties", "dead"])
def test_flash_attention_merge_edge_cases(ca, kind):
    dev = ca.HipTensorDevice(0)
    rng = np.random.default_rng(7)
    for seq in (130, 513, 3000):
        err = run_case(ca, dev, rng, 32, 8, 128, seq, 32, spread=4.0, kind=kind)
        assert err <= 2e-5, (kind, seq, err)


def f64_causal_attention(q, k, v, n_heads, n_kv, hd, pos0, rows):
    """row r (position pos0 + r) over the cached positions 0 .. pos0 + r; the kernel rounds the probabilities to f16 before
    the second product (as the reference does, batch_matmul.rs:39): restated here, so that the bound measures the kernel"""
    q16 = q.astype(np.float16).astype(np.float64).reshape(rows, n_heads, hd)
    kf = k.view(np.float16).astype(np.float64).reshape(n_kv, pos0 + rows, hd)
    vf = v.view(np.float16).astype(np.float64).reshape(n_kv, pos0 + rows, hd)
    grp = n_heads // n_kv
    out = np.zeros((rows, n_heads, hd))
    for h in range(n_heads):
        s = q16[:, h, :] @ kf[h // grp].T                      # rows x positions
        mask = np.arange(pos0 + rows)[None, :] > (pos0 + np.arange(rows))[:, None]
        s[mask] = -np.inf
        p = np.exp(s - s.max(axis=1, keepdims=True))
        out[:, h, :] = (p / p.sum(axis=1, keepdims=True)) @ vf[h // grp]
    return out.reshape(-1)


@pytest.mark.parametrize("n_heads,n_kv,hd", [(32, 8, 128), (8, 8, 128), (8, 1, 128), (8, 2, 64), (4, 4, 64)])
def test_flash_prefill_attention_equals_float64_arithmetic(ca, n_heads, n_kv, hd):
    """k_attn_flash_rows -- the fast prompt pass's causal attention on the f16 matrix cores -- against float64 softmax(q K^T) V
    on the same f16 inputs: whole and ragged 64-row workgroups, batches that start in the middle of the cache (pos0 > 0), tiles
    of 64 cached positions with and without a tail.  The probabilities enter the second matrix product as f16 (2^-12 relative
    each, as in the reference): stated bound 1.5e-3 of max|out| (observed ~3e-4)."""
    dev = ca.HipTensorDevice(0)
    rng = np.random.default_rng(31 * n_kv + hd)
    worst = 0.0
    for pos0, rows in ((0, 1), (0, 17), (0, 64), (0, 65), (0, 200), (0, 512), (5, 33), (64, 64), (100, 129), (777, 300)):
        q = (rng.standard_normal(rows * n_heads * hd) / np.sqrt(hd)).astype(np.float32)
        k = rng.standard_normal((n_kv, pos0 + rows, hd)).astype(np.float16)
        v = rng.standard_normal((n_kv, pos0 + rows, hd)).astype(np.float16)
        got = dev.debug_flash_attention_rows(q, k.view(np.uint16).reshape(-1), v.view(np.uint16).reshape(-1), n_heads, n_kv, hd, pos0, rows)
        ref = f64_causal_attention(q, k.view(np.uint16), v.view(np.uint16), n_heads, n_kv, hd, pos0, rows)
        # the cache behind the live positions holds whatever an earlier sequence or the allocator left there: NaN / Inf / huge
        # values in the tail of the K / V buffers must not reach any output (masked scores; zero probabilities times V)
        for junk in (0xFFFF, 0x7C00, 0x7BFF):
            cap = pos0 + rows + 37
            kj = np.full((n_kv, cap, hd), junk, dtype=np.uint16)
            vj = np.full((n_kv, cap, hd), junk, dtype=np.uint16)
            kj[:, :pos0 + rows] = k.view(np.uint16)
            vj[:, :pos0 + rows] = v.view(np.uint16)
            gj = dev.debug_flash_attention_rows(q, kj.reshape(-1), vj.reshape(-1), n_heads, n_kv, hd, pos0, rows)
            assert np.array_equal(gj.view(np.uint32), got.view(np.uint32)), (pos0, rows, hex(junk))
        assert np.all(np.isfinite(got)), (pos0, rows)
        err = float(np.max(np.abs(got - ref)) / np.max(np.abs(ref)))
        worst = max(worst, err)
        assert err <= 1.5e-3, (pos0, rows, err)
    print(f"flash prefill vs float64, {n_heads} heads / {n_kv} kv x {hd}: worst {worst:.2e} of max|out|")


def test_a_cache_length_that_is_no_multiple_of_eight_keeps_the_split_kv_kernels(ca):
    """The exact long-context kernels read score / probability rows of seq_len elements as 16-byte vectors (seq_len % 8 == 0);
    k_attn_flash reads cache rows only.  A cache of 203 positions used to drop the fast step to one workgroup per head past 96
    positions (45 us per layer at 900 positions of the 8B shape); now it runs the same kernels as a cache of 208 -- the cache row
    stride differs, nothing else: the logits agree bit for bit."""
    model = synth.build_model(synth.SHAPES["tiny-hd128"], synth.Q4_0, seed=77)
    dev = ca.HipTensorDevice(0)
    conf, w = synth.to_hip(model, dev)
    odd = ca.HipLlamaRunner(conf, w, dev, 203, True)
    even = ca.HipLlamaRunner(conf, w, dev, 208, True)
    odd_eager = ca.HipLlamaRunner(conf, w, dev, 203, True, False)
    for i in range(200):
        t = (5 * i + 2) % 1000
        a, b, c = odd.forward(t, i).copy(), even.forward(t, i).copy(), odd_eager.forward(t, i).copy()
        assert np.array_equal(a.view(np.uint32), b.view(np.uint32)), f"cache of 203 vs 208 positions, step {i}"
        assert np.array_equal(a.view(np.uint32), c.view(np.uint32)), f"graph vs eager, step {i}"

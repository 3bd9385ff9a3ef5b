"""GPU parity for the engine (crabml_amd/csrc/engine.hpp): wo + RMSNorm + gate/up + SiLU*mul + ffn_down + RMSNorm of a layer as
ONE persistent launch fed by an LDS-DMA loader over a CU-major copy of the weights.

The engine restates the arithmetic of the three launches it replaces (k_gemv_res_nq, k_gateup_q, k_gemv_res_nq) in the same
summation orders, so the test is an equality test: the logits of every step, hipGraph replay or eager launches, are
bit-identical to the 5-launch layer's -- at shapes that exercise every partition rule (16 rows per CU / two CUs per norm chunk,
CUs without rows, one and two gate/up blocks per CU, several row counts per slot, ragged ring depths) -- and, through the oracle,
to the reference: strict-order device == oracle is asserted elsewhere, here fast + engine stays inside the pinned tolerance and
equals the fast 5-launch path exactly (matmul_vec.rs:26-78, rms_norm.rs:33-46, silu.rs:6-13, buf_q8_0.rs:87-134)."""
import os

import numpy as np
import pytest

from crabml_amd import synth
from oracle import oracle as o
from tests.helpers import check_fast, to_oracle

pytestmark = pytest.mark.gpu
ENGINE = 524288  # CRABML_HIP_LLAMA_ENGINE
TOKS = [1, 365, 400, 282, 7, 9, 11, 13, 21, 34]

SHAPES = {
    # dim / 16 = 32 CUs with rows, 32 gate/up blocks: every CU has rows and one block
    "tiny-gqa": synth.SHAPES["tiny-gqa"],
    # 48 gate/up blocks over 48 CUs, only 32 of them own wo / down rows
    "rows<cus": synth.ModelShape("rows<cus", 512, 1536, 2, 8, 2, 1024, 64, 1e-5, None),
    # 288 blocks over 256 CUs: CUs 0..31 carry two gate/up blocks; 224 CUs own no rows at all
    "two-blocks": synth.ModelShape("two-blocks", 512, 9216, 2, 8, 4, 1024, 64, 1e-5, None),
    # head_dim 48 (the attention launch does not quantize its output: a quantizer launch feeds the engine), 18 CUs with rows, 24 blocks
    "15m": synth.SHAPES["15m"],
    # long ffn_down rows: 1152 blocks = 20736 bytes per row -> does not fit a slot -> the engine must decline, not break
    "row>slot": synth.ModelShape("row>slot", 256, 36864, 1, 4, 4, 512, 32, 1e-5, None),
}


def run(ca, model, flags, toks, use_graph=True, seq_len=64):
    dev = ca.HipTensorDevice(0)
    conf, w = synth.to_hip(model, dev)
    r = ca.HipLlamaRunner(conf, w, dev, seq_len, True, use_graph, True, extra_flags=flags)
    return [r.forward(t, i).copy() for i, t in enumerate(toks)], r


@pytest.mark.parametrize("shape", ["tiny-gqa", "rows<cus", "two-blocks", "15m", "row>slot"])
def test_engine_equals_the_five_launch_layer(ca, shape):
    model = synth.build_model(SHAPES[shape], synth.TYPE_BY_NAME["Q4_0"], seed=31)
    base, _ = run(ca, model, 0, TOKS)
    for use_graph in (True, False):
        got, _ = run(ca, model, ENGINE, TOKS, use_graph)
        for i, (a, b) in enumerate(zip(got, base)):
            assert np.array_equal(a.view(np.uint32), b.view(np.uint32)), f"{shape} graph={use_graph} step {i}"


@pytest.mark.parametrize("nc,depth,thin", [(1, 3, 0), (2, 4, 1), (3, 5, 2), (7, 0, 0), (15, 0, 3)])
def test_engine_wave_counts_and_ring_depths(ca, nc, depth, thin):
    """1..15 consumer waves, shallow rings (a slot is reused after 3 fills), the thinned / paused loader: same bits."""
    model = synth.build_model(SHAPES["two-blocks"], synth.TYPE_BY_NAME["Q4_0"], seed=32)
    base, _ = run(ca, model, 0, TOKS[:6])
    env = {"CRABML_HIP_ENGINE_NC": str(nc), "CRABML_HIP_ENGINE_D": str(depth), "CRABML_HIP_ENGINE_THIN": str(thin)}
    old = {k: os.environ.get(k) for k in env}
    os.environ.update(env)
    try:
        got, _ = run(ca, model, ENGINE, TOKS[:6])
    finally:
        for k, v in old.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v
    for i, (a, b) in enumerate(zip(got, base)):
        assert np.array_equal(a.view(np.uint32), b.view(np.uint32)), f"nc={nc} D={depth} thin={thin} step {i}"


def test_engine_against_the_oracle(ca):
    model = synth.build_model(SHAPES["tiny-gqa"], synth.TYPE_BY_NAME["Q4_0"], seed=33)
    odev = o.OracleDevice(thread_num=4)
    oconf, ow = to_oracle(model, odev)
    orr = o.OracleLlamaRunner(oconf, ow, odev, 64, True)
    ref = [orr.forward([t], i).copy() for i, t in enumerate(TOKS)]
    got, r = run(ca, model, ENGINE, TOKS)
    err = np.array([np.max(np.abs(x - y)) / np.max(np.abs(y)) for x, y in zip(got, ref)])
    check_fast("engine/tiny-gqa/Q4_0", "Q4_0", err)
    ids = r.decode_greedy(int(o.argmax_last(got[-1])), 8)
    assert len(ids) == 8 and r.kv_cache_len() == len(TOKS) + 8


def test_engine_llama3_8b_shape_two_layers(ca):
    """The headline shape (dim 4096, hidden 14336, 32 / 8 heads): 256 CUs x 16 rows, 448 gate/up blocks (192 CUs carry two),
    8 / 8 / 2 rows per slot, ring of 7 slots; 40 decode steps through the graph, every logit compared."""
    model = synth.build_model(synth.SHAPES["llama3-8b"], synth.TYPE_BY_NAME["Q4_0"], seed=34, n_layers=2)
    toks = [(37 * i + 11) % 128256 for i in range(40)]
    base, _ = run(ca, model, 0, toks, seq_len=128)
    got, _ = run(ca, model, ENGINE, toks, seq_len=128)
    for i, (a, b) in enumerate(zip(got, base)):
        assert np.array_equal(a.view(np.uint32), b.view(np.uint32)), f"step {i}"


def test_engine_after_a_batched_prefill(ca):
    """prompt through crabml_hip_llama_prefill (the engine plays no part in it), then decode steps through the engine: tokens and
    logits equal the 5-launch step's, and a reset + second sequence reuses the same weight stream."""
    model = synth.build_model(SHAPES["rows<cus"], synth.TYPE_BY_NAME["Q4_0"], seed=35)
    prompt = [(11 * i + 3) % 1024 for i in range(24)]
    out = []
    for flags in (0, ENGINE):
        dev = ca.HipTensorDevice(0)
        conf, w = synth.to_hip(model, dev)
        r = ca.HipLlamaRunner(conf, w, dev, 64, True, True, True, extra_flags=flags)
        first = r.prefill(prompt).copy()
        steps = [r.forward(t, len(prompt) + i).copy() for i, t in enumerate(TOKS[:5])]
        ids = list(r.decode_greedy(7, 6))
        r.reset()
        again = r.prefill(prompt[:9]).copy()
        out.append((first, steps, ids, again))
    a, b = out
    assert np.array_equal(a[0].view(np.uint32), b[0].view(np.uint32))
    for x, y in zip(a[1], b[1]):
        assert np.array_equal(x.view(np.uint32), y.view(np.uint32))
    assert a[2] == b[2]
    assert np.array_equal(a[3].view(np.uint32), b[3].view(np.uint32))

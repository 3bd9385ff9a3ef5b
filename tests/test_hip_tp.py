"""GPU parity of the tensor-parallel decode step (crabml_hip_llama_config_t.tp_*, crabml_amd/csrc/fused.hip).

A 1-GPU box cannot host an RCCL group of several ranks, so the tp ranks are created on ONE device without a
communicator and driven by crabml_hip_llama_tp_sim_forward: same kernels, same shards, same partial-sum
hand-off, the all-reduce replaced by a rank-order sum.  On a strict-order device that is bit-identical to the
oracle's tensor-parallel restatement; on the fast device it meets the single-GPU tolerance.  The RCCL
binding itself (dlopen, unique id, communicator, all-reduce on the device stream) is exercised with a
1-rank communicator."""
import numpy as np
import pytest

from crabml_amd import synth, tp as tp_mod
from oracle import oracle as o
from tests.helpers import to_oracle

pytestmark = pytest.mark.gpu
TOKS = [1, 365, 400, 282, 7, 9]


def oracle_tp_logits(model, tp, kv_f16, toks):
    odev = o.OracleDevice(thread_num=4)
    rank_w = []
    for r in range(tp):
        conf, w = to_oracle(tp_mod.shard_model(model, tp, r, kv_f16), odev)
        rank_w.append(w)
    runner = o.OracleTpLlamaRunner(conf, rank_w, odev, 64, kv_f16)
    return [runner.forward([t], i).copy() for i, t in enumerate(toks)]


def hip_tp_ranks(ca, model, tp, kv_f16, dev):
    ranks = []
    for r in range(tp):
        conf, w = synth.to_hip(tp_mod.shard_model(model, tp, r, kv_f16), dev)
        ranks.append(ca.HipLlamaRunner(conf, w, dev, 64, kv_f16, True, True, tp, r))
    return ranks


@pytest.mark.parametrize("shape,tp,kv_f16,fmt", [("tiny-gqa", 2, True, "Q4_0"), ("tiny-gqa", 2, True, "Q8_0"),
                                                 ("15m", 3, False, "Q8_0"), ("15m", 3, True, "Q4_0"),
                                                 ("tiny-gqa", 2, True, "Q4_K"), ("tiny-gqa", 2, True, "Q4_1")])
def test_tp_sim_strict_is_bit_exact(ca, shape, tp, kv_f16, fmt):
    model = synth.build_model(synth.SHAPES[shape], synth.TYPE_BY_NAME[fmt], seed=31)
    ref = oracle_tp_logits(model, tp, kv_f16, TOKS)
    dev = ca.HipTensorDevice(0, False, 0, True)
    ranks = hip_tp_ranks(ca, model, tp, kv_f16, dev)
    for i, t in enumerate(TOKS):
        lg = ca.HipLlamaRunner.tp_sim_forward(ranks, t, i)
        assert np.array_equal(lg.view(np.uint32), ref[i].view(np.uint32)), f"step {i}"
    assert all(r.kv_cache_len() == len(TOKS) for r in ranks)


@pytest.mark.parametrize("fmt", ["Q4_0", "Q8_0", "Q4_K", "Q4_1", "Q6_K"])
def test_tp_sim_fast_meets_single_gpu_tolerance(ca, fmt):
    model = synth.build_model(synth.SHAPES["tiny-gqa"], synth.TYPE_BY_NAME[fmt], seed=32)
    odev = o.OracleDevice(thread_num=4)
    oconf, ow = to_oracle(model, odev)
    orr = o.OracleLlamaRunner(oconf, ow, odev, 64, True)
    ref = [orr.forward([t], i).copy() for i, t in enumerate(TOKS)]
    dev = ca.HipTensorDevice(0)
    ranks = hip_tp_ranks(ca, model, 2, True, dev)
    got = [ca.HipLlamaRunner.tp_sim_forward(ranks, t, i).copy() for i, t in enumerate(TOKS)]
    err = np.array([np.max(np.abs(a - b)) / np.max(np.abs(b)) for a, b in zip(got, ref)])
    assert err[0] <= 2e-2 and np.median(err) <= 3e-2 and np.max(err) <= 1e-1, err


def test_tp_rank_needs_the_group_driver_and_matching_shards(ca):
    model = synth.build_model(synth.SHAPES["tiny-gqa"], synth.Q4_0, seed=33)
    dev = ca.HipTensorDevice(0)
    ranks = hip_tp_ranks(ca, model, 2, True, dev)
    with pytest.raises(ca.CrabmlError):  # a lone rank cannot step: its partial sums are never reduced
        ranks[0].forward(1, 0)
    with pytest.raises(ca.CrabmlError):  # wrong order = wrong tp_rank
        ca.HipLlamaRunner.tp_sim_forward(ranks[::-1], 1, 0)
    conf, w = synth.to_hip(model, dev)  # unsharded weights do not fit a tp = 2 rank
    with pytest.raises(ca.CrabmlError):
        ca.HipLlamaRunner(conf, w, dev, 64, True, True, True, 2, 0)


def test_rccl_binding_single_rank_all_reduce(ca):
    dev = ca.HipTensorDevice(0)
    uid = ca.TpComm.unique_id()
    assert len(uid) == 128
    comm = ca.TpComm(dev, uid, 1, 0)
    assert (comm.nranks, comm.rank) == (1, 0)
    x = np.arange(4096, dtype=np.float32) - 7.5
    t = ca.HipTensor.from_cpu(x.view(np.uint8), [4096], ca.GGMLType.F32, dev)
    comm.all_reduce(t)  # sum over one rank: identity, but goes through ncclAllReduce on the device stream
    assert np.array_equal(t.export(), x)


@pytest.mark.parametrize("fmt,tp,strict", [("Q4_0", 2, True), ("Q4_0", 2, False), ("Q4_K", 2, True), ("Q8_0", 2, False)])
def test_tp_sim_with_the_classifier_split_by_vocabulary(ca, fmt, tp, strict):
    """output.weight split by rows, per-shard arg-max (last maximum, sampler.rs:109-116), {max, index} pairs
    combined in rank order.  Logits (gathered shard by shard) equal the replicated classifier's bit for bit -- every row dot is
    the same dot -- and on the strict device the oracle's; the sampled tokens equal the replicated run's."""
    model = synth.build_model(synth.SHAPES["tiny-gqa"], synth.TYPE_BY_NAME[fmt], seed=36)
    dev = ca.HipTensorDevice(0, False, 0, strict)
    rep = hip_tp_ranks(ca, model, tp, True, dev)
    split = []
    for r in range(tp):
        conf, w = synth.to_hip(tp_mod.shard_model(model, tp, r, True, split_vocab=True), dev)
        split.append(ca.HipLlamaRunner(conf, w, dev, 64, True, True, True, tp, r, extra_flags=tp_mod.TP_SPLIT_VOCAB))
    ref = None
    if strict:
        odev = o.OracleDevice(thread_num=4)
        rank_w = [to_oracle(tp_mod.shard_model(model, tp, r, True, split_vocab=True), odev) for r in range(tp)]
        orr = o.OracleTpLlamaRunner(rank_w[0][0], [w for _, w in rank_w], odev, 64, True)
        ref = [orr.forward([t], i).copy() for i, t in enumerate(TOKS)]
    for i, t in enumerate(TOKS):
        a = ca.HipLlamaRunner.tp_sim_forward(rep, t, i)
        b = ca.HipLlamaRunner.tp_sim_forward(split, t, i)
        assert np.array_equal(a.view(np.uint32), b.view(np.uint32)), f"step {i}"
        if ref is not None:
            assert np.array_equal(b.view(np.uint32), ref[i].view(np.uint32)), f"oracle, step {i}"


def test_the_vocabulary_split_is_validated_at_create(ca):
    model = synth.build_model(synth.SHAPES["tiny-gqa"], synth.Q4_0, seed=37)
    dev = ca.HipTensorDevice(0)
    conf, w = synth.to_hip(tp_mod.shard_model(model, 2, 0, True), dev)  # replicated classifier, flag set: wrong row count
    with pytest.raises(ca.CrabmlError):
        ca.HipLlamaRunner(conf, w, dev, 64, True, True, True, 2, 0, extra_flags=tp_mod.TP_SPLIT_VOCAB)
    conf, w = synth.to_hip(tp_mod.shard_model(model, 2, 0, True, split_vocab=True), dev)  # split shard without the flag
    with pytest.raises(ca.CrabmlError):
        ca.HipLlamaRunner(conf, w, dev, 64, True, True, True, 2, 0)

"""CPU tests of the tensor-parallel shard planner (crabml_amd/tp.py) against the oracle.

  * shards are byte slices that re-assemble to the full tensors;
  * the oracle's tensor-parallel restatement with tp = 1 IS the reference runner (bit for bit), and with
    tp > 1 agrees with it up to f32 re-association of the two row-parallel GEMVs per layer;
  * the unique-id hand-off of init_tp_comm over a 2-process gloo group."""
import numpy as np
import pytest

from crabml_amd import synth
from oracle import oracle as o
from tests.helpers import to_oracle

tp_mod = pytest.importorskip("crabml_amd.tp")


def test_shards_reassemble():
    model = synth.build_model(synth.SHAPES["tiny-gqa"], synth.Q4_0, seed=21)
    shards = [tp_mod.shard_model(model, 2, r) for r in range(2)]
    for name, t in model.tensors.items():
        parts = [s.tensors[name] for s in shards]
        rows = t.shape[0]
        if name.endswith(("attn_q.weight", "attn_k.weight", "attn_v.weight", "ffn_gate.weight", "ffn_up.weight")):
            assert np.array_equal(np.concatenate([p.data for p in parts]), t.data), name
            assert sum(p.shape[0] for p in parts) == rows
        elif name.endswith(("attn_output.weight", "ffn_down.weight")):
            full = np.concatenate([p.data.reshape(rows, -1) for p in parts], axis=1)
            assert np.array_equal(full.reshape(-1), t.data), name
            assert sum(p.shape[1] for p in parts) == t.shape[1]
        else:
            assert all(p is t for p in parts), name


@pytest.mark.parametrize("shape,tp,wt,kv16", [
    ("15m", 2, synth.Q4_0, True),        # 3 heads x 48 = 144 columns: not a multiple of the 32-element block
    ("tiny-gqa", 4, synth.Q4_0, True),   # 2 kv heads cannot feed 4 ranks
    ("tiny-gqa", 2, synth.Q4_0, False),  # GQA over the f32 cache pairs head h with kv head h % n_kv
    ("llama3-8b", 16, synth.Q4_0, True),  # more than one node
])
def test_invalid_splits_are_rejected(shape, tp, wt, kv16):
    with pytest.raises(ValueError):
        tp_mod.check_tp(synth.SHAPES[shape], tp, wt, kv16)


def test_valid_splits():
    tp_mod.check_tp(synth.SHAPES["tiny-gqa"], 2, synth.Q4_K, True)  # k slices of 256 and 512: whole super-blocks
    tp_mod.check_tp(synth.SHAPES["llama3-70b"], 8, synth.Q4_0, True)  # BASELINE config C5
    tp_mod.check_tp(synth.SHAPES["15m"], 3, synth.Q8_0, False)  # MHA: the f32 cache shards too
    assert tp_mod.allreduce_bytes_per_token(synth.SHAPES["llama3-70b"], 8) == 2 * 80 * 8192 * 4


def _tp_logits(model, tp, kv_f16, toks):
    odev = o.OracleDevice(thread_num=2)
    rank_w = []
    for r in range(tp):
        conf, w = to_oracle(tp_mod.shard_model(model, tp, r, kv_f16), odev)
        rank_w.append(w)
    runner = o.OracleTpLlamaRunner(conf, rank_w, odev, 64, kv_f16)
    return [runner.forward([t], i).copy() for i, t in enumerate(toks)]


@pytest.mark.parametrize("shape,tp,kv_f16", [("tiny-gqa", 2, True), ("15m", 3, False)])
def test_oracle_tp_matches_unsharded_reference(shape, tp, kv_f16):
    model = synth.build_model(synth.SHAPES[shape], synth.Q8_0, seed=22, n_layers=2)
    toks = [1, 365, 400, 7]
    odev = o.OracleDevice(thread_num=2)
    conf, w = to_oracle(model, odev)
    ref_runner = o.OracleLlamaRunner(conf, w, odev, 64, kv_f16)
    ref = [ref_runner.forward([t], i).copy() for i, t in enumerate(toks)]
    one = _tp_logits(model, 1, kv_f16, toks)
    for a, b in zip(one, ref):  # tp = 1: the very same op sequence
        assert np.array_equal(a.view(np.uint32), b.view(np.uint32))
    got = _tp_logits(model, tp, kv_f16, toks)
    # splitting k only re-associates f32 sums; the truncating Q8_0 activation quantizer can amplify that
    # (the same bound the reference's own scalar-vs-AVX2 paths need, tests/test_hip_fused.py)
    err = np.array([np.max(np.abs(a - b)) / np.max(np.abs(b)) for a, b in zip(got, ref)])
    assert err[0] <= 2e-2 and np.median(err) <= 3e-2 and np.max(err) <= 1e-1, err


def _uid_worker(rank, world, port, q):
    import torch.distributed as dist

    dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{port}", rank=rank, world_size=world)
    try:
        seen = {}

        class FakeCa:  # stands in for the GPU-only TpComm: records what init_tp_comm hands over
            class TpComm:
                @staticmethod
                def unique_id():
                    return bytes(range(128))

                def __init__(self, device, uid, nranks, rk):
                    seen.update(uid=uid, nranks=nranks, rank=rk)

        import sys
        real = sys.modules.get("crabml_amd")
        proxy = type(sys)("crabml_amd")
        proxy.__dict__.update(real.__dict__)
        proxy.TpComm = FakeCa.TpComm
        sys.modules["crabml_amd"] = proxy
        try:
            tp_mod.init_tp_comm(None, rank, world, tp_mod.torch_broadcast(rank))
        finally:
            sys.modules["crabml_amd"] = real
        q.put((rank, seen["uid"] == bytes(range(128)), seen["nranks"], seen["rank"]))
    finally:
        dist.destroy_process_group()


def test_unique_id_broadcast_over_gloo():
    import torch.multiprocessing as mp
    import socket

    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_uid_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in procs)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert res == [(0, True, 2, 0), (1, True, 2, 1)]

"""Tensor-parallel shard planner for the fused decode step (BASELINE config C5).

Megatron-style split of the per-layer GEMVs over `tp` ranks, one process per GPU:

  column-parallel (rows of W):   attn_q / attn_k / attn_v by heads, ffn_gate / ffn_up by rows
  row-parallel    (k of W):      attn_output by the local heads' columns, ffn_down by the local hidden columns
  replicated:                    token_embd, norms; output (classifier) unless split_vocab
  split_vocab:    output.weight by rows -- rank r scores vocabulary [r V / tp, (r + 1) V / tp), takes the arg-max of
                                 its shard and the ranks exchange 8-byte {max, index} pairs (CRABML_HIP_LLAMA_TP_SPLIT_VOCAB)

Both splits cut GGML block rows on block boundaries (k_local % block_elems == 0), so a shard is a plain
byte slice of the GGUF tensor: no re-quantization, and the activation blocks each rank quantizes are the
very blocks the unsharded model quantizes.  The two dim-sized partial sums per layer are all-reduced
(RCCL over xGMI: 2 x dim x 4 bytes per layer per token); see crabml_hip_llama_config_t.tp_*.

Pure numpy host logic -- never imports oracle/.
"""
from __future__ import annotations

from typing import Optional

import numpy as np

from .synth import BLOCK_BYTES, BLOCK_ELEMS, RawModel, RawTensor


def check_tp(shape, tp: int, wtype: int, kv_f16: bool = True) -> None:
    """Raises ValueError unless `shape` can be split over `tp` ranks (mirrors crabml_hip_llama_create)."""
    if tp < 1 or tp > 8:
        raise ValueError(f"tp={tp}: 1..8 ranks (one xGMI-connected node)")
    if shape.n_kv_heads % tp or shape.n_heads % tp or shape.hidden % tp:
        raise ValueError(f"tp={tp} must divide n_heads={shape.n_heads}, n_kv_heads={shape.n_kv_heads} and hidden={shape.hidden}")
    be = max(BLOCK_ELEMS[wtype], 32)
    dim_l = shape.n_heads // tp * shape.head_dim
    if dim_l % be or (shape.hidden // tp) % be:
        raise ValueError(f"tp={tp}: local k slices ({dim_l}, {shape.hidden // tp}) must be multiples of the block size {be}")
    if tp > 1 and not kv_f16 and shape.n_heads != shape.n_kv_heads:
        raise ValueError("tensor-parallel GQA needs the f16 kv cache (the f32 cache pairs head h with kv head h % n_kv)")


def _rows(t: RawTensor, lo: int, hi: int) -> RawTensor:
    rows, cols = t.shape
    rb = cols // BLOCK_ELEMS[t.typ] * BLOCK_BYTES[t.typ]
    return RawTensor(np.ascontiguousarray(t.data.reshape(rows, rb)[lo:hi]).reshape(-1), [hi - lo, cols], t.typ)


def _cols(t: RawTensor, lo: int, hi: int) -> RawTensor:
    rows, cols = t.shape
    be, bb = BLOCK_ELEMS[t.typ], BLOCK_BYTES[t.typ]
    assert lo % be == 0 and hi % be == 0
    blocks = t.data.reshape(rows, cols // be, bb)
    return RawTensor(np.ascontiguousarray(blocks[:, lo // be:hi // be]).reshape(-1), [rows, hi - lo], t.typ)


TP_SPLIT_VOCAB = 1048576  # CRABML_HIP_LLAMA_TP_SPLIT_VOCAB (include/crabml_hip.h)


def shard_model(model: RawModel, tp: int, rank: int, kv_f16: bool = True, split_vocab: bool = False) -> RawModel:
    """Rank `rank`'s shard of `model`.  `.shape` stays the GLOBAL ModelShape (the runner derives the local
    geometry from tp_size); the sharded tensors carry their local [rows, cols].  split_vocab: output.weight is cut by rows
    (the runner must then be created with the TP_SPLIT_VOCAB flag)."""
    s = model.shape
    check_tp(s, tp, model.wtype, kv_f16)
    if not 0 <= rank < tp:
        raise ValueError(f"rank {rank} outside 0..{tp - 1}")
    if tp == 1:
        return model
    if split_vocab and (s.vocab % tp or "output.weight" not in model.tensors):
        raise ValueError(f"split_vocab: vocab {s.vocab} must be a multiple of tp={tp} and output.weight must not be tied")
    v_lo, v_hi = rank * (s.vocab // tp), (rank + 1) * (s.vocab // tp)
    hd = s.head_dim
    q_lo, q_hi = rank * (s.n_heads // tp) * hd, (rank + 1) * (s.n_heads // tp) * hd
    kv_lo, kv_hi = rank * (s.n_kv_heads // tp) * hd, (rank + 1) * (s.n_kv_heads // tp) * hd
    h_lo, h_hi = rank * (s.hidden // tp), (rank + 1) * (s.hidden // tp)
    out = RawModel(s, model.wtype)
    for name, t in model.tensors.items():
        if name.endswith("attn_q.weight"):
            out.tensors[name] = _rows(t, q_lo, q_hi)
        elif name.endswith("attn_k.weight") or name.endswith("attn_v.weight"):
            out.tensors[name] = _rows(t, kv_lo, kv_hi)
        elif name.endswith("attn_output.weight"):
            out.tensors[name] = _cols(t, q_lo, q_hi)
        elif name.endswith("ffn_gate.weight") or name.endswith("ffn_up.weight"):
            out.tensors[name] = _rows(t, h_lo, h_hi)
        elif name.endswith("ffn_down.weight"):
            out.tensors[name] = _cols(t, h_lo, h_hi)
        elif name == "output.weight" and split_vocab:
            out.tensors[name] = _rows(t, v_lo, v_hi)
        else:
            out.tensors[name] = t  # replicated
    return out


def allreduce_bytes_per_token(shape, tp: int) -> int:
    """Bytes each rank contributes to all-reduces per decoded token (2 per layer, dim f32 each)."""
    return 0 if tp <= 1 else 2 * shape.n_layers * shape.dim * 4


def init_tp_comm(device, rank: int, world: int, broadcast_bytes):
    """Create this rank's RCCL communicator.  `broadcast_bytes(b: Optional[bytes]) -> bytes` ships rank 0's
    128-byte ncclUniqueId to every rank over any side channel (torch.distributed gloo/nccl object broadcast,
    a file, MPI ...): rank 0 passes the id, the others pass None."""
    import crabml_amd as ca

    uid: Optional[bytes] = ca.TpComm.unique_id() if rank == 0 else None
    uid = broadcast_bytes(uid)
    if not isinstance(uid, (bytes, bytearray)) or len(uid) != 128:
        raise ValueError("init_tp_comm: the broadcast did not deliver a 128-byte ncclUniqueId")
    return ca.TpComm(device, bytes(uid), world, rank)


def init_tp_p2p(device, rank: int, world: int, max_elems: int, all_gather_bytes):
    """Create this rank's one-shot P2P all-reduce group (crabml_hip_tp_p2p_*: the production collective, no RCCL on the
    data path).  `all_gather_bytes(b: bytes) -> list[bytes]` returns every rank's 64-byte inbox handle in rank order over
    any side channel (torch.distributed all_gather_object, files, MPI ...)."""
    import crabml_amd as ca

    comm = ca.TpComm.p2p(device, world, rank, max_elems)
    if world > 1:
        handles = all_gather_bytes(comm.export_handle())
        if len(handles) != world or any(len(h) != 64 for h in handles):
            raise ValueError("init_tp_p2p: the all-gather did not deliver one 64-byte handle per rank")
        comm.connect(b"".join(bytes(h) for h in handles))
    return comm


def torch_all_gather(world: int):
    """all_gather_bytes for init_tp_p2p over an initialised torch.distributed process group."""
    import torch.distributed as dist

    def gather(b):
        box = [None] * world
        dist.all_gather_object(box, b)
        return box

    return gather


def file_all_gather(directory: str, rank: int, world: int, timeout_s: float = 120.0):
    """all_gather_bytes over a shared directory (no torch): every rank drops its handle as a file and waits for the rest."""
    import os
    import time

    def gather(b):
        tmp = os.path.join(directory, f"handle.{rank}.tmp")
        with open(tmp, "wb") as f:
            f.write(b)
        os.replace(tmp, os.path.join(directory, f"handle.{rank}"))
        out, t0 = [], time.time()
        for r in range(world):
            path = os.path.join(directory, f"handle.{r}")
            while not os.path.exists(path):
                if time.time() - t0 > timeout_s:
                    raise TimeoutError(f"rank {rank}: no handle from rank {r}")
                time.sleep(0.01)
            with open(path, "rb") as f:
                out.append(f.read())
        return out

    return gather


def torch_broadcast(rank: int):
    """broadcast_bytes for init_tp_comm over an initialised torch.distributed process group."""
    import torch.distributed as dist

    def bcast(b):
        box = [b]
        dist.broadcast_object_list(box, src=0)
        return box[0]

    return bcast

"""Synthetic GGUF-layout weights for the bench and the parity tests.

There is no network and the reference's quantized fixtures are stripped, so models are filled directly in
the GGML block byte layout (crabml-core/src/cpu/buf/buf_q*.rs): block scales d = f16(U(2e-3, 2e-2)),
quants uniform over their full range.  Pure numpy -- this module never touches oracle/.
The same bytes feed the HIP backend (HipTensor.from_cpu) and, in tests / the CPU baseline, the oracle.
"""
from __future__ import annotations

from dataclasses import dataclass, field
from typing import Dict, List, Optional

import numpy as np

F32, F16, Q4_0, Q4_1, Q8_0, Q8_1, Q4_K, Q5_K, Q6_K, Q8_K = 0, 1, 2, 3, 8, 9, 12, 13, 14, 15
Q5_0, Q5_1, Q2_K, Q3_K = 6, 7, 10, 11
BLOCK_ELEMS = {F32: 1, F16: 1, Q4_0: 32, Q4_1: 32, Q8_0: 32, Q8_1: 32, Q4_K: 256, Q5_K: 256, Q6_K: 256, Q8_K: 256, Q5_0: 32, Q5_1: 32, Q2_K: 256, Q3_K: 256}
BLOCK_BYTES = {F32: 4, F16: 2, Q4_0: 18, Q4_1: 20, Q8_0: 34, Q8_1: 36, Q4_K: 144, Q5_K: 176, Q6_K: 210, Q8_K: 292, Q5_0: 22, Q5_1: 24, Q2_K: 84, Q3_K: 110}
TYPE_NAMES = {F32: "F32", F16: "F16", Q4_0: "Q4_0", Q4_1: "Q4_1", Q8_0: "Q8_0", Q4_K: "Q4_K", Q5_K: "Q5_K", Q6_K: "Q6_K", Q8_K: "Q8_K",
              Q5_0: "Q5_0", Q5_1: "Q5_1", Q2_K: "Q2_K", Q3_K: "Q3_K"}
TYPE_BY_NAME = {v: k for k, v in TYPE_NAMES.items()}


def _f16_scales(rng, n, lo=2e-3, hi=2e-2):
    """n positive f16 scales in [lo, hi], drawn uniformly over the f16 BIT patterns of that range
    (log-uniform in value) -- an integer draw, ~10x faster than uniform-float + astype(float16)."""
    b0 = int(np.array([lo], dtype=np.float16).view(np.uint16)[0])
    b1 = int(np.array([hi], dtype=np.float16).view(np.uint16)[0])
    return rng.integers(b0, b1 + 1, size=n, dtype=np.uint16)


_POOL = {}


def _rand_bytes(rng: np.random.Generator, n: int) -> np.ndarray:
    """n pseudo-random bytes at memcpy speed: a 32 MiB PCG64 pool, re-entered at a random offset per call
    (multi-GB models would otherwise spend a minute in the generator; HBM traffic does not care that the
    byte stream repeats every 32 MiB at different addresses)."""
    entry = _POOL.get("pool")
    if entry is None or entry[0] is not rng:  # identity, not id(): a freed generator's id can be reused
        pool = np.frombuffer(rng.bytes(32 << 20), dtype=np.uint8)
        _POOL["pool"] = (rng, pool)
    else:
        pool = entry[1]
    if n <= 4096:
        return np.frombuffer(rng.bytes(n), dtype=np.uint8).copy()
    off = int(rng.integers(0, pool.size))
    out = np.empty(n, dtype=np.uint8)
    pos = 0
    while pos < n:
        take = min(n - pos, pool.size - off)
        out[pos:pos + take] = pool[off:off + take]
        pos += take
        off = 0
    return out


# Q4_1 blocks with m drawn independently of d (rounds 1-3): a large common offset in every GEMV, which makes relative logit errors
# small -- the fast kernels were pinned at (1.5e-3, 2e-3) on these weights, and one test keeps that pin (tests/test_hip_fused.py)
Q4_1_INDEPENDENT_M = False


def random_blocks(rng: np.random.Generator, n_elems: int, typ: int, scale_mul: float = 1.0) -> np.ndarray:
    """Raw bytes (uint8) of n_elems elements in GGML layout `typ`, filled with random quants."""
    be, bb = BLOCK_ELEMS[typ], BLOCK_BYTES[typ]
    assert n_elems % be == 0
    nb = n_elems // be
    if typ == F32:
        return (rng.standard_normal(n_elems, dtype=np.float32) * np.float32(0.02 * scale_mul)).view(np.uint8)
    if typ == F16:
        return (rng.standard_normal(n_elems, dtype=np.float32) * np.float32(0.02 * scale_mul)).astype(np.float16).view(np.uint8)
    lo, hi = 2e-3 * scale_mul, 2e-2 * scale_mul
    if typ in (Q4_0, Q8_0):  # all bytes random, then the d field overwritten
        out = _rand_bytes(rng, nb * bb).reshape(nb, bb)
        b0, b1 = (lo, hi) if typ == Q4_0 else (lo / 8, hi / 8)
        lo16 = int(np.array([b0], dtype=np.float16).view(np.uint16)[0])
        span = int(np.array([b1], dtype=np.float16).view(np.uint16)[0]) - lo16 + 1
        sc = (lo16 + _rand_bytes(rng, nb * 2).view(np.uint16) % span).astype(np.uint16)
        out[:, 0:2] = sc.reshape(nb, 1).view(np.uint8)
        return out.reshape(-1)
    out = np.empty((nb, bb), dtype=np.uint8)
    if typ == Q4_1:
        # m centres the block's levels 0..15 (m ~ -7.5 d): weights with a common offset make a 32-layer model's residual stream grow
        # until the f16 KV cache overflows (NaN logits at the 8B depth with the independent m of rounds 1-3)
        d16 = _f16_scales(rng, nb, lo, hi)
        out[:, 0:2] = d16.reshape(nb, 1).view(np.uint8)
        if Q4_1_INDEPENDENT_M:
            m = -rng.uniform(8 * lo, 8 * hi, size=nb)
        else:
            m = -7.5 * d16.view(np.float16).astype(np.float32) * rng.uniform(0.9, 1.1, size=nb).astype(np.float32)
        out[:, 2:4] = m.astype(np.float16).view(np.uint16).reshape(nb, 1).view(np.uint8)
        out[:, 4:] = rng.integers(0, 256, size=(nb, 16), dtype=np.uint8)
    elif typ == Q4_K:
        out[:, 0:2] = _f16_scales(rng, nb, lo / 32, hi / 32).reshape(nb, 1).view(np.uint8)
        out[:, 2:4] = _f16_scales(rng, nb, lo / 4, hi / 4).reshape(nb, 1).view(np.uint8)
        out[:, 4:] = rng.integers(0, 256, size=(nb, 140), dtype=np.uint8)  # 6-bit scales/mins + nibbles
    elif typ == Q5_K:  # the REFERENCE's order: qs[128] | qh[32] | scales[12] | d f16 | dmin f16 (buf_q5_k.rs:13-21)
        out[:, 0:172] = rng.integers(0, 256, size=(nb, 172), dtype=np.uint8)
        out[:, 172:174] = _f16_scales(rng, nb, lo / 64, hi / 64).reshape(nb, 1).view(np.uint8)
        out[:, 174:176] = _f16_scales(rng, nb, lo / 4, hi / 4).reshape(nb, 1).view(np.uint8)
    elif typ == Q6_K:  # ql[128] | qh[64] | scales i8[16] | d f16 (buf_q6_k.rs:11-18)
        out[:, 0:192] = rng.integers(0, 256, size=(nb, 192), dtype=np.uint8)
        out[:, 192:208] = rng.integers(-64, 64, size=(nb, 16), dtype=np.int8).view(np.uint8)
        out[:, 208:210] = _f16_scales(rng, nb, lo / 256, hi / 256).reshape(nb, 1).view(np.uint8)
    elif typ == Q5_0:  # d f16 | qh[4] | qs[16] (buf_q5_0.rs:13-19): levels -16..15
        out[:, 0:2] = _f16_scales(rng, nb, lo / 2, hi / 2).reshape(nb, 1).view(np.uint8)
        out[:, 2:] = rng.integers(0, 256, size=(nb, 20), dtype=np.uint8)
    elif typ == Q5_1:  # d f16 | m f16 | qh[4] | qs[16] (buf_q5_1.rs:10-17): levels 0..31, m centres them
        d16 = _f16_scales(rng, nb, lo / 2, hi / 2)
        out[:, 0:2] = d16.reshape(nb, 1).view(np.uint8)
        m = -15.5 * d16.view(np.float16).astype(np.float32) * rng.uniform(0.9, 1.1, size=nb).astype(np.float32)
        out[:, 2:4] = m.astype(np.float16).view(np.uint16).reshape(nb, 1).view(np.uint8)
        out[:, 4:] = rng.integers(0, 256, size=(nb, 20), dtype=np.uint8)
    elif typ == Q2_K:  # scales[16] (scale | min << 4) | qs[64] | d f16 | dmin f16 (buf_q2_k.rs:17-28)
        out[:, 0:80] = rng.integers(0, 256, size=(nb, 80), dtype=np.uint8)
        out[:, 80:82] = _f16_scales(rng, nb, lo / 4, hi / 4).reshape(nb, 1).view(np.uint8)
        out[:, 82:84] = _f16_scales(rng, nb, lo / 3, hi / 3).reshape(nb, 1).view(np.uint8)
    elif typ == Q3_K:  # hmask[32] | qs[64] | scales[12] | d f16 (buf_q3_k.rs:19-30): 6-bit scales - 32, levels -4..3
        out[:, 0:108] = rng.integers(0, 256, size=(nb, 108), dtype=np.uint8)
        out[:, 108:110] = _f16_scales(rng, nb, lo / 8, hi / 8).reshape(nb, 1).view(np.uint8)
    elif typ == Q8_K:
        out[:, 0:4] = rng.uniform(lo / 8, hi / 8, size=nb).astype(np.float32).reshape(nb, 1).view(np.uint8)
        q = rng.integers(-127, 128, size=(nb, 256), dtype=np.int8)
        out[:, 4:260] = q.view(np.uint8)
        out[:, 260:] = q.reshape(nb, 16, 16).astype(np.int16).sum(axis=2).astype(np.int16).view(np.uint8).reshape(nb, 32)
    else:
        raise ValueError(f"unsupported type {typ}")
    return out.reshape(-1)


@dataclass
class ModelShape:
    name: str
    dim: int
    hidden: int
    n_layers: int
    n_heads: int
    n_kv_heads: int
    vocab: int
    seq_len: int
    rms_eps: float = 1e-5
    rope_dim: Optional[int] = None

    @property
    def head_dim(self):
        return self.dim // self.n_heads

    @property
    def kv_dim(self):
        return self.dim * self.n_kv_heads // self.n_heads


# shapes of the configs
SHAPES = {
    "15m": ModelShape("tinyllamas-stories-15m", 288, 768, 6, 6, 6, 32000, 256, 1e-5, 48),
    "llama3-8b": ModelShape("Llama-3-8B", 4096, 14336, 32, 32, 8, 128256, 8192, 1e-5, None),
    "llama3-70b": ModelShape("Llama-3-70B", 8192, 28672, 80, 64, 8, 128256, 8192, 1e-5, None),
    # a small GQA shape with every dim a multiple of 256 (valid for all formats incl. K-quants); used by tests
    "tiny-gqa": ModelShape("tiny-gqa", 512, 1024, 2, 8, 2, 1024, 64, 1e-5, None),
    # ... and one with Llama-3's head_dim of 128 (the kernels specialized for it: k_attn_s<128>, k_attn_wo)
    "tiny-hd128": ModelShape("tiny-hd128", 512, 1024, 2, 4, 2, 1024, 128, 1e-5, None),
}


@dataclass
class RawTensor:
    data: np.ndarray  # uint8 raw GGML bytes
    shape: List[int]
    typ: int


@dataclass
class RawModel:
    shape: ModelShape
    wtype: int
    tensors: Dict[str, RawTensor] = field(default_factory=dict)

    def gemv_weight_bytes_per_token(self) -> int:
        """Algorithmic weight bytes one decode step streams through matmul_vec."""
        s = self.shape
        total = 0
        for name, t in self.tensors.items():
            if name.endswith("_norm.weight") or name == "token_embd.weight":
                continue
            n = 1
            for d in t.shape:
                n *= d
            total += n // BLOCK_ELEMS[t.typ] * BLOCK_BYTES[t.typ]
        if "output.weight" not in self.tensors:  # tied classifier
            t = self.tensors["token_embd.weight"]
            total += s.vocab * s.dim // BLOCK_ELEMS[t.typ] * BLOCK_BYTES[t.typ]
        return total


def use_more_bits(i_layer: int, n_layers: int) -> bool:
    """llama.cpp's rule for the layers whose attn_v / ffn_down tensors get the wider type in the *_K_M mixes."""
    return i_layer < n_layers // 8 or i_layer >= 7 * n_layers // 8 or (i_layer - n_layers // 8) % 3 == 2


def flip_scale_signs(model: "RawModel", seed: int = 5) -> None:
    """Q4_0 blocks with d of EITHER sign, in place (what a quantizer that divides by the signed maximum produces,
    buf_q4_0.rs:96-104).  random_blocks draws every d > 0, i.e. a mean level of -0.5 d: a common-mode component in every GEMV,
    which a metric relative to max|logit| rewards.  Zero-mean weights are the hard case for any re-association / re-rounding
    (profiles/r04_reference_order_sensitivity.log: the reference's own scalar and AVX2 builds differ by 7-10 % of max|logit| there)."""
    rng = np.random.default_rng(seed)
    for t in model.tensors.values():
        if t.typ == Q4_0:
            blk = t.data.reshape(-1, 18)
            blk[:, 1] ^= (rng.integers(0, 2, size=blk.shape[0], dtype=np.uint8) << 7)


def build_model(shape: ModelShape, wtype: int, seed: int = 8, n_layers: Optional[int] = None,
                embed_type: Optional[int] = None, tp: int = 1, output_type: Optional[int] = None,
                k_m_mix: bool = False, tp_split_vocab: bool = False) -> RawModel:
    """All-`wtype` synthetic Llama weights with GGUF tensor names (model.rs:228-283); norms are F32
    (the loader dequantizes them, model.rs:267-282).  tp > 1: the tensors get one rank's LOCAL shard shapes
    (what crabml_amd.tp.shard_model would cut; random bytes either way -- for timing one rank of a large model
    without materialising all of it).  k_m_mix (with wtype = Q4_K): the tensor-type recipe of llama.cpp's Q4_K_M
    files -- attn_v and ffn_down in Q6_K on the `use_more_bits` layers, output.weight in Q6_K -- i.e. different
    GGML types inside one layer (all with the Q8_K rhs)."""
    rng = np.random.default_rng(seed)
    L = shape.n_layers if n_layers is None else n_layers
    shp = ModelShape(**{**shape.__dict__, "n_layers": L})
    m = RawModel(shp, wtype)
    et = wtype if embed_type is None else embed_type

    def add(name, rows, cols, typ, scale_mul=1.0):
        m.tensors[name] = RawTensor(random_blocks(rng, rows * cols, typ, scale_mul), [rows, cols], typ)

    def norm(name, n):
        w = (1.0 + rng.standard_normal(n) * 0.01).astype(np.float32)
        m.tensors[name] = RawTensor(w.view(np.uint8), [n], F32)

    add("token_embd.weight", shape.vocab, shape.dim, et, 4.0)
    dim_l, kv_l, hid_l = shape.dim // tp, shape.kv_dim // tp, shape.hidden // tp
    for l in range(L):
        add(f"blk.{l}.attn_q.weight", dim_l, shape.dim, wtype)
        add(f"blk.{l}.attn_k.weight", kv_l, shape.dim, wtype)
        wide = Q6_K if (k_m_mix and use_more_bits(l, L)) else wtype
        add(f"blk.{l}.attn_v.weight", kv_l, shape.dim, wide)
        add(f"blk.{l}.attn_output.weight", shape.dim, dim_l, wtype)
        add(f"blk.{l}.ffn_gate.weight", hid_l, shape.dim, wtype)
        add(f"blk.{l}.ffn_down.weight", shape.dim, hid_l, wide)
        add(f"blk.{l}.ffn_up.weight", hid_l, shape.dim, wtype)
        norm(f"blk.{l}.attn_norm.weight", shape.dim)
        norm(f"blk.{l}.ffn_norm.weight", shape.dim)
    norm("output_norm.weight", shape.dim)
    # llama.cpp's "Q4_0" / "Q4_K_M" files keep output.weight in Q6_K: `output_type` builds that mix
    # tp_split_vocab (with tp > 1): one rank's vocabulary shard of the classifier (CRABML_HIP_LLAMA_TP_SPLIT_VOCAB)
    add("output.weight", shape.vocab // tp if tp_split_vocab else shape.vocab, shape.dim,
        (Q6_K if k_m_mix else wtype) if output_type is None else output_type)
    return m


def to_hip(model: RawModel, device):
    """Upload a RawModel through Tensor::from_cpu -> (LlamaConfig, LlamaWeights) of the hip backend."""
    import crabml_amd as ca

    tmap = {F32: ca.GGMLType.F32, F16: ca.GGMLType.F16, Q4_0: ca.GGMLType.Q4_0, Q4_1: ca.GGMLType.Q4_1,
            Q8_0: ca.GGMLType.Q8_0, Q4_K: ca.GGMLType.Q4K, Q5_K: ca.GGMLType.Q5K, Q6_K: ca.GGMLType.Q6K, Q8_K: ca.GGMLType.Q8K,
            Q5_0: ca.GGMLType.Q5_0, Q5_1: ca.GGMLType.Q5_1, Q2_K: ca.GGMLType.Q2K, Q3_K: ca.GGMLType.Q3K}
    s = model.shape

    def up(name):
        t = model.tensors[name]
        return ca.HipTensor.from_cpu(t.data, t.shape, tmap[t.typ], device)

    w = ca.LlamaWeights()
    w.token_embed = up("token_embd.weight")
    w.wq = [up(f"blk.{l}.attn_q.weight") for l in range(s.n_layers)]
    w.wk = [up(f"blk.{l}.attn_k.weight") for l in range(s.n_layers)]
    w.wv = [up(f"blk.{l}.attn_v.weight") for l in range(s.n_layers)]
    w.wo = [up(f"blk.{l}.attn_output.weight") for l in range(s.n_layers)]
    w.ffn_gate_weight = [up(f"blk.{l}.ffn_gate.weight") for l in range(s.n_layers)]
    w.ffn_down_weight = [up(f"blk.{l}.ffn_down.weight") for l in range(s.n_layers)]
    w.ffn_up_weight = [up(f"blk.{l}.ffn_up.weight") for l in range(s.n_layers)]
    w.rms_att_weight = [up(f"blk.{l}.attn_norm.weight") for l in range(s.n_layers)]
    w.rms_ffn_weight = [up(f"blk.{l}.ffn_norm.weight") for l in range(s.n_layers)]
    w.rms_final_weight = up("output_norm.weight")
    if "output.weight" in model.tensors:
        w.output_weight = up("output.weight")
    conf = ca.LlamaConfig(embedding_dim=s.dim, hidden_dim=s.hidden, n_layers=s.n_layers, n_heads=s.n_heads,
                          n_kv_heads=s.n_kv_heads, vocab_size=s.vocab, seq_len=s.seq_len, rms_norm_eps=s.rms_eps,
                          rope_dim=s.rope_dim)
    return conf, w


# ---- GGUF container writer (test / tool side of crabml_amd/csrc/host/gguf.hpp) ----------------------------------------
# Layout per crabml-core/src/gguf.rs:499-566 (header + metadata), :632-646 (tensor infos), :722-724 (data alignment):
# lengths are u32 in v1 and u64 in v2 / v3; tensor dims are stored innermost-first (model.rs:473-475 reverses them).
_GGUF_T = {"u8": 0, "i8": 1, "u16": 2, "i16": 3, "u32": 4, "i32": 5, "f32": 6, "bool": 7, "str": 8, "arr": 9,
           "u64": 10, "i64": 11, "f64": 12}
_GGUF_FMT = {"u8": "<B", "i8": "<b", "u16": "<H", "i16": "<h", "u32": "<I", "i32": "<i", "f32": "<f", "bool": "<B",
             "u64": "<Q", "i64": "<q", "f64": "<d"}


def write_gguf(model: RawModel, path: str, version: int = 3, alignment: int = 32, write_alignment_key=None,
               extra_kv=None, tensor_order=None, data_start: str = "reference", pad_header_to_alignment: bool = False) -> None:
    """Serialize a RawModel as a llama-architecture GGUF file.  write_alignment_key: None = omit general.alignment
    (readers assume 32), or a value-type name ("u32", "u64", "i32", ...) to store `alignment` under that type.
    extra_kv: list of (key, type_name, value); arrays as (key, "arr", (elem_type_name, [values])).
    data_start: "reference" = always skip to the NEXT multiple of the alignment (gguf.rs:722-724: a whole extra block when
    the tensor infos already end aligned), "spec" = pad only when misaligned (llama.cpp / gguf-py writers).  The two differ
    only when the header ends on a boundary; pad_header_to_alignment forces that case (a filler string key sized so)."""
    import struct

    s = model.shape

    def wlen(n):
        return struct.pack("<I" if version == 1 else "<Q", n)

    def wstr(x):
        b = x.encode("utf-8")
        return wlen(len(b)) + b

    def wval(t, v):
        if t == "str":
            return wstr(v)
        if t == "arr":
            et, items = v
            return struct.pack("<I", _GGUF_T[et]) + wlen(len(items)) + b"".join(wval(et, i) for i in items)
        return struct.pack(_GGUF_FMT[t], v)

    kv = [("general.architecture", "str", "llama"), ("general.name", "str", s.name),
          ("llama.context_length", "u32", s.seq_len), ("llama.embedding_length", "u32", s.dim),
          ("llama.block_count", "u32", s.n_layers), ("llama.feed_forward_length", "u32", s.hidden),
          ("llama.attention.head_count", "u32", s.n_heads), ("llama.attention.head_count_kv", "u32", s.n_kv_heads),
          ("llama.attention.layer_norm_rms_epsilon", "f32", s.rms_eps)]
    if s.rope_dim is not None:
        kv.append(("llama.rope.dimension_count", "u32", s.rope_dim))
    kv += [("tokenizer.ggml.model", "str", "llama"),
           ("tokenizer.ggml.tokens", "arr", ("str", [f"<{i}>" for i in range(s.vocab)])),
           ("tokenizer.ggml.bos_token_id", "u32", 1), ("tokenizer.ggml.eos_token_id", "u32", 2)]
    if write_alignment_key is not None:
        kv.append(("general.alignment", write_alignment_key, alignment))
    kv += list(extra_kv or [])
    names = tensor_order or list(model.tensors.keys())

    def header(kvs):
        h = struct.pack("<II", 0x46554747, version) + wlen(len(names)) + wlen(len(kvs))
        for k, t, v in kvs:
            h += wstr(k) + struct.pack("<I", _GGUF_T[t]) + wval(t, v)
        return h

    def infos_len():
        n = 0
        for nm in names:
            t = model.tensors[nm]
            n += len(wstr(nm)) + 4 + len(t.shape) * (4 if version == 1 else 8) + 12
        return n

    if pad_header_to_alignment:  # a filler key whose string length makes header + tensor infos end on a boundary
        base = len(header(kv + [("x.filler", "str", "")])) + infos_len()
        kv.append(("x.filler", "str", "." * ((alignment - base % alignment) % alignment)))
    out = header(kv)
    infos = b""
    off = 0
    offsets = []
    for n in names:
        t = model.tensors[n]
        offsets.append(off)
        infos += wstr(n) + struct.pack("<I", len(t.shape))
        for d in reversed(t.shape):
            infos += struct.pack("<I" if version == 1 else "<Q", d)
        infos += struct.pack("<IQ", t.typ, off)
        off += (len(t.data) + alignment - 1) // alignment * alignment
    out += infos
    pos = len(out)
    if data_start == "spec":
        pad = (alignment - pos % alignment) % alignment
    else:
        pad = pos - (pos % alignment) + alignment - pos  # gguf.rs:722-724: a whole extra block when already aligned
    if pad_header_to_alignment:
        assert pos % alignment == 0, pos
    with open(path, "wb") as f:
        f.write(out)
        f.write(b"\0" * pad)
        for n, o in zip(names, offsets):
            t = model.tensors[n]
            f.write(t.data.tobytes())
            f.write(b"\0" * ((len(t.data) + alignment - 1) // alignment * alignment - len(t.data)))


def load_gguf_hip(path: str, device):
    """(LlamaConfig, LlamaWeights<HipTensor>) of a llama GGUF file through the C++ loader (gguf.hpp)."""
    import crabml_amd as ca

    gf = ca.GGUFFile(path)
    conf = gf.load_config()
    return conf, gf.load_weights(conf, device)

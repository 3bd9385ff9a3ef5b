"""Shared helpers of the parity tests (test infrastructure: may import oracle/)."""
import numpy as np

from crabml_amd import synth
from oracle import oracle as o


def to_oracle(model: synth.RawModel, odev):
    """RawModel -> (oracle LlamaConfig, LlamaWeights of OracleTensor) -- what CpuLlamaModelLoader builds."""
    s = model.shape

    def up(name):
        t = model.tensors[name]
        return o.OracleTensor.from_bytes(t.data, t.typ, t.shape, odev)

    w = o.LlamaWeights()
    w.token_embed = up("token_embd.weight")
    for l in range(s.n_layers):
        w.wq.append(up(f"blk.{l}.attn_q.weight"))
        w.wk.append(up(f"blk.{l}.attn_k.weight"))
        w.wv.append(up(f"blk.{l}.attn_v.weight"))
        w.wo.append(up(f"blk.{l}.attn_output.weight"))
        w.ffn_gate_weight.append(up(f"blk.{l}.ffn_gate.weight"))
        w.ffn_down_weight.append(up(f"blk.{l}.ffn_down.weight"))
        w.ffn_up_weight.append(up(f"blk.{l}.ffn_up.weight"))
        w.rms_att_weight.append(up(f"blk.{l}.attn_norm.weight"))
        w.rms_ffn_weight.append(up(f"blk.{l}.ffn_norm.weight"))
    w.rms_final_weight = up("output_norm.weight")
    if "output.weight" in model.tensors:
        w.output_weight = up("output.weight")
    conf = o.LlamaConfig(s.dim, s.hidden, s.n_layers, s.n_heads, s.n_kv_heads, s.vocab, s.seq_len, s.rms_eps,
                         s.rope_dim)
    return conf, w


def gemv_order_bound(w_raw, wtyp, x, m, k):
    """|sum_b t_b computed in any order - any other order| <= (n-1) eps sum|t_b| <= this bound.
    Uses the dequantized values: sum_i |w_i||x_i| >= sum_b |t_b| (x is the f32 activation; its
    quantization error is common to both sides)."""
    wd = o.dequantize(w_raw, wtyp).reshape(m, k).astype(np.float64)
    if wtyp == o.Q4_1:  # the reference's Q4_1 dequantize is interleaved; magnitudes are all we need here
        pass
    return np.abs(wd) @ np.abs(x.astype(np.float64))


# CRABML_HIP_LLAMA_EXACT_NORM: the fast step keeps RMSNorm's division in the producing launch (in-launch gather) -- the form whose
# variants (own launch / epilogue / split chunks) are bit-identical to each other; the default hands 1 / rms to the consumer
EXACT_NORM = 8388608

GEMV_REL = 2e-5  # f32 re-association bound factor: |gpu - oracle| <= GEMV_REL * sum_i |w_i x_i| + tiny

# End-to-end tolerance of the FAST kernels PER WEIGHT FORMAT: (median, max) over the tested steps of
# max|hip - oracle| / max|oracle logit|, teacher-forced on the oracle's tokens.  Set to a small multiple (~3x) of what
# MI355X shows on the test seeds; every fast-path test records what it observed in gpurun_out/fast_path_errors.json
# (tests/golden/fast_path_errors_observed.json keeps the last committed copy).  The formats with a TRUNCATING rhs quantizer
# (Q8_0 / Q8_1 rhs: buf_q8_0.rs:119-124) amplify 1-ulp GEMV differences into +-1 quant flips -- the reference's own
# scalar-vs-AVX2 spread on Q8_0 models is 1.5-2.7e-2 (tests/test_oracle_runner.py) --, the K-quants round to nearest
# (buf_q8_k.rs:84-131), F32 / F16 weights have no activation quantizer at all.
# Observed on MI355X (round 2, tests/golden/fast_path_errors_observed.json), worst (median, max) over all test models:
#   Q4_0 3.0e-3 / 7.5e-3   Q8_0 2.7e-2 / 3.7e-2   Q4_1 4.1e-4 / 5.1e-4   Q4_K, Q6_K, Q8_K 1.9e-7 / 2.8e-7
#   F32 1.1e-4 / 2.6e-4    F16 4.1e-4 / 5.6e-4     (full 8B shape, Q4_0 and Q4_K: 6e-5 / 5e-4, tests/test_hip_headline.py)
# K-quants: the rhs quantizer rounds to nearest, so a step normally reproduces the oracle to ~2e-7 (median bound 1e-6) -- but the
# re-associated GEMV sums still move a value across a rounding boundary now and then: 55 of 2100 decode steps of the 1-layer
# tiny-gqa model show ONE flipped quant (5e-3 .. 1.1e-2 of max|logit|; round 4, profiles/r04_kquant_flip_rate.log), with the
# exact and the one-workgroup attention kernels alike.  The max bound has to admit such a step.  On the 2-layer tiny-gqa model
# (twice the quantization sites, one more layer to amplify a flip) most 24-step runs contain one and the largest seen is 3.75e-2
# (same log, second table): the K-quant max bound is 2 x that.  The MEDIAN bound (5e-7) is what pins the kernels; at the 8B shape a
# flipped quant moves the logits by < 5e-4 (tests/test_hip_headline.py).
# Round 4: every bound is at most 2 x the largest value observed on MI355X for that format (tests/golden/
# fast_path_errors_observed.json; the K-quant max is 2 x the largest single-flip step).
# Round 6: the K-quant max admits ONE flipped quant of the size seen on the 2-layer test model (largest: 3.75e-2; bound 1.2 x that,
# it was 2 x) -- the flip-COUNT gate below is what keeps a change that corrupts many steps from hiding under it; Q8_0's bounds are
# per model where the models differ (FAST_TOL_MODEL: 2 x observed on each).
FAST_TOL = {"Q4_0": (8e-3, 1.5e-2), "Q8_0": (6.2e-2, 7.5e-2), "Q4_1": (3e-2, 4e-2), "Q4_K": (5e-7, 4.5e-2), "Q5_K": (5e-7, 4.5e-2), "Q6_K": (5e-7, 4.5e-2),
            "Q8_K": (5e-7, 4.5e-2), "F32": (2.5e-4, 5.5e-4), "F16": (8.5e-4, 1.2e-3),
            # the scalar-only formats (round 4), 2 x observed: Q5_0 shares the truncating Q8_0 rhs quantizer and shows Q8_0-sized flips
            # on the tiny-gqa model (median 1.2e-2 / 1.9e-2, max 1.5e-2 / 2.5e-2 on the trait / graph path; Q8_0: 1.5e-2 / 1.8e-2) while
            # its single GEMVs sit inside the re-association bound (tests/test_hip_gemv.py) and the strict step is bit-exact;
            # Q5_1 behaves like Q4_1 (Q8_1 rhs), Q2_K / Q3_K like the other K-quants.
            # Q4_1 / Q5_1 (round 4, late): the synthetic blocks now centre their levels (m ~ -7.5 d / -15.5 d; with the independent m of
            # rounds 1-3 the 8B-depth model overflowed the f16 KV cache).  Zero-mean weights lose the common-mode component that made
            # the relative error look small (4e-4): observed now 1.2-1.5e-2 median, 1.5-1.8e-2 max -- the size of the reference's OWN
            # scalar-vs-AVX2 spread on zero-mean models (profiles/r04_reference_order_sensitivity.log)
            "Q5_0": (4e-2, 5e-2), "Q5_1": (3e-2, 4e-2), "Q2_K": (5e-7, 4.5e-2), "Q3_K": (5e-7, 4.5e-2)}
# per (model, format) where a model shows less than the format's worst: 2 x observed on MI355X (gpurun_out/fast_path_errors.json of
# round 6: tiny-gqa Q8_0 median 1.8e-2 / max 2.4e-2 over the trait and fused paths; the 15m model is the format row: 3.1e-2 / 3.7e-2)
FAST_TOL_MODEL = {("tiny-gqa", "Q8_0"): (3.6e-2, 4.7e-2)}
# The fast step's long-context attention (k_attn_flash: f32 exp / f32 accumulation instead of the reference's f16 exp table, f16
# probabilities, f16 products and serial f16 sum) against the oracle, positions 224 .. 4095: (median, max) bounds, 2 x observed.
# With a round-to-nearest rhs quantizer (K-quants) the 1e-4-sized attention deviation flips a quant of wo's rhs at most steps:
# the K-quant row is a flip-sized bound, not the 2e-7 of the exact kernels.
FAST_TOL.update({"FLASH:Q4_0": (8.6e-3, 1.2e-2), "FLASH:Q4_K": (7e-3, 3e-2), "FLASH:Q8_0": (5.5e-2, 7.5e-2), "FLASH:F32": (5e-3, 1e-2)})
_OBSERVED = {}


def check_fast(key, fmt, err):
    """records the observed per-step errors under `key` and asserts the format's tolerance"""
    import json
    import os

    err = np.asarray(err, dtype=np.float64)
    _OBSERVED[key] = {"median": float(np.median(err)), "max": float(np.max(err)), "steps": int(err.size)}
    try:
        os.makedirs("gpurun_out", exist_ok=True)
        path = os.path.join("gpurun_out", "fast_path_errors.json")
        prev = {}
        if os.path.exists(path):
            try:
                prev = json.load(open(path))
            except ValueError:
                prev = {}
        prev.update(_OBSERVED)
        with open(path, "w") as f:
            json.dump(prev, f, indent=1, sort_keys=True)
    except OSError:
        pass
    med, mx = FAST_TOL[fmt]
    parts = key.split("/")
    if len(parts) > 1 and (parts[1], fmt) in FAST_TOL_MODEL:
        med, mx = FAST_TOL_MODEL[(parts[1], fmt)]
    assert np.median(err) <= med and np.max(err) <= mx, (key, err)
    if fmt in ROUND_TO_NEAREST:
        # a second gate for the formats whose rhs quantizer rounds to nearest: the max bound has to admit a step with ONE flipped
        # quant (above), but flips are rare events -- a change that corrupts many steps by a flip-sized amount must not hide under it
        flips = int(np.sum(err > FLIP_SIZED))
        assert flips <= max(3, err.size // 4), (key, "flip-sized steps", flips, err)


# K-quants: the rhs quantizer (buf_q8_k.rs:84-131) rounds half away from zero; a step without a flipped quant sits at ~2e-7
ROUND_TO_NEAREST = ("Q4_K", "Q5_K", "Q6_K", "Q8_K", "Q2_K", "Q3_K")
FLIP_SIZED = 1e-4


# ---- an independent (pure Python) GGUF reader for the checker side ------------------------------------------------------
# The product parses GGUF in C++ (crabml_amd/csrc/host/gguf.hpp); the oracle side of a real-file test must not lean on
# it, so this is a second, minimal reader written from the format description in crabml-core/src/gguf.rs:499-566
# (header + metadata), :632-646 (tensor infos) and the GGUF spec's data-section rule (pad to the alignment only when
# misaligned).  Returns a synth.RawModel whose tensors are views of the file's bytes.
def read_gguf_py(path):
    import struct

    buf = open(path, "rb").read()
    pos = 0

    def rd(fmt):
        nonlocal pos
        v = struct.unpack_from(fmt, buf, pos)
        pos += struct.calcsize(fmt)
        return v[0]

    magic, version = rd("<I"), rd("<I")
    assert magic == 0x46554747 and version in (1, 2, 3)
    ln = (lambda: rd("<I")) if version == 1 else (lambda: rd("<Q"))

    def rstr():
        nonlocal pos
        n = ln()
        s = buf[pos:pos + n]
        pos += n
        return s.decode("utf-8", "replace")

    scalar = {0: "<B", 1: "<b", 2: "<H", 3: "<h", 4: "<I", 5: "<i", 6: "<f", 7: "<B", 10: "<Q", 11: "<q", 12: "<d"}

    def rval(t):
        if t == 8:
            return rstr()
        if t == 9:
            et = rd("<I")
            return [rval(et) for _ in range(ln())]
        return rd(scalar[t])

    n_tensors, n_kv = ln(), ln()
    kv = {}
    for _ in range(n_kv):
        k = rstr()
        kv[k] = rval(rd("<I"))
    infos = []
    for _ in range(n_tensors):
        name = rstr()
        nd = rd("<I")
        dims = [ln() for _ in range(nd)]
        typ, off = rd("<I"), rd("<Q")
        infos.append((name, dims[::-1], typ, off))  # stored innermost-first (model.rs:473-475 reverses them)
    al = int(kv.get("general.alignment", 32))
    start = pos + (al - pos % al) % al
    arch = kv["general.architecture"]
    shape = synth.ModelShape(kv.get("general.name", "gguf"), kv[f"{arch}.embedding_length"], kv[f"{arch}.feed_forward_length"],
                             kv[f"{arch}.block_count"], kv[f"{arch}.attention.head_count"], kv[f"{arch}.attention.head_count_kv"],
                             len(kv["tokenizer.ggml.tokens"]), kv[f"{arch}.context_length"],
                             kv[f"{arch}.attention.layer_norm_rms_epsilon"], kv.get(f"{arch}.rope.dimension_count"))
    model = synth.RawModel(shape, infos[1][2])
    raw = np.frombuffer(buf, dtype=np.uint8)
    for name, dims, typ, off in infos:
        n = int(np.prod(dims))
        nbytes = n // synth.BLOCK_ELEMS[typ] * synth.BLOCK_BYTES[typ]
        model.tensors[name] = synth.RawTensor(raw[start + off:start + off + nbytes], dims, typ)
    return model, kv

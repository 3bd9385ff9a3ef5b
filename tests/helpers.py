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


GEMV_REL = 2e-5  # f32 re-association bound factor: |gpu - oracle| <= GEMV_REL * sum_i |w_i x_i| + tiny

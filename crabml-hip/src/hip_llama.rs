//! `HipLlamaRunner`: the decode step of `Llama2Runner<T>` for the Llama architecture behind
//! `crabml_hip_llama_*` -- the same op sequence (crabml-llama2/src/llama2.rs:184-281, 527-638) as fused kernels
//! replayed from one hipGraph, with the greedy sampler (sampler.rs:109-116) on the device, plus the prompt pass of
//! `Llama2Runner::prefill` (llama2.rs:111-129) as batched passes on the matrix cores.
//!
//! The generic `Llama2Runner<HipTensor>` needs none of this (every `Tensor` method is a kernel launch); it is
//! launch-bound on an MI355X (about 1000 launches per token for Llama-3-8B), which is what this path removes.
//! Architectures other than Llama (gemma, qwen2, phi2: llama2.rs:283-524) stay on the generic runner.

use std::ptr;
use std::sync::Arc;

use crabml::bail;
use crabml::error::ErrorKind;
use crabml::error::Result;
use crabml::tensor::Tensor;
use crabml_llama2::model::LlamaConfig;
use crabml_llama2::model::LlamaWeights;
use crabml_llama2::model::ModelArchitecture;

use crate::ffi;
use crate::HipTensor;
use crate::HipTensorDeviceRef;

pub struct HipLlamaRunner {
    raw: *mut ffi::crabml_hip_llama_t,
    conf: LlamaConfig,
    device: HipTensorDeviceRef,
    // the context holds its own references on the weight buffers; the Arc keeps the Rust view of them alive, too
    _weights: Arc<LlamaWeights<HipTensor>>,
    logits: Vec<f32>,
}

unsafe impl Send for HipLlamaRunner {}

fn handles(ts: &[HipTensor]) -> Vec<*const ffi::crabml_hip_buf_t> {
    ts.iter().map(|t| t.raw()).collect()
}

impl HipLlamaRunner {
    /// Counterpart of `Llama2Runner::new` (llama2.rs:46-100): `seq_len` is the KV-cache capacity.
    /// Fails with `NotImplemented` for weights the fused step does not take (tensor types with different rhs formats
    /// inside one layer, biases, fused wqkv); the caller then falls back to `Llama2Runner<HipTensor>`.
    pub fn new(
        conf: &LlamaConfig,
        weights: Arc<LlamaWeights<HipTensor>>,
        device: HipTensorDeviceRef,
        seq_len: usize,
        use_f16_kv_cache: bool,
    ) -> Result<Self> {
        if conf.architecture != ModelArchitecture::Llama {
            bail!(
                ErrorKind::NotImplemented,
                "the fused hip decode step serves the llama architecture only, got {:?}",
                conf.architecture
            );
        }
        let w = &weights;
        if !w.wqkv.is_empty()
            || !w.bq.is_empty()
            || !w.bk.is_empty()
            || !w.bv.is_empty()
            || !w.bo.is_empty()
            || !w.bqkv.is_empty()
            || !w.ffn_down_bias.is_empty()
            || !w.ffn_up_bias.is_empty()
            || !w.rms_att_bias.is_empty()
            || w.rms_final_bias.is_some()
        {
            bail!(
                ErrorKind::NotImplemented,
                "the fused hip decode step takes no biases / fused qkv weights"
            );
        }
        let n = conf.n_layers;
        for (name, v) in [
            ("wq", &w.wq),
            ("wk", &w.wk),
            ("wv", &w.wv),
            ("wo", &w.wo),
            ("ffn_gate_weight", &w.ffn_gate_weight),
            ("ffn_down_weight", &w.ffn_down_weight),
            ("ffn_up_weight", &w.ffn_up_weight),
            ("rms_att_weight", &w.rms_att_weight),
            ("rms_ffn_weight", &w.rms_ffn_weight),
        ] {
            if v.len() != n {
                bail!(
                    ErrorKind::ModelError,
                    "{} holds {} tensors for {} layers",
                    name,
                    v.len(),
                    n
                );
            }
        }
        let (rms_att, rms_ffn) = (handles(&w.rms_att_weight), handles(&w.rms_ffn_weight));
        let (wq, wk, wv, wo) = (handles(&w.wq), handles(&w.wk), handles(&w.wv), handles(&w.wo));
        let (gate, down, up) = (
            handles(&w.ffn_gate_weight),
            handles(&w.ffn_down_weight),
            handles(&w.ffn_up_weight),
        );
        let c_weights = ffi::crabml_hip_llama_weights_t {
            token_embed: w.token_embed.raw(),
            rms_att_weight: rms_att.as_ptr(),
            rms_ffn_weight: rms_ffn.as_ptr(),
            wq: wq.as_ptr(),
            wk: wk.as_ptr(),
            wv: wv.as_ptr(),
            wo: wo.as_ptr(),
            ffn_gate_weight: gate.as_ptr(),
            ffn_down_weight: down.as_ptr(),
            ffn_up_weight: up.as_ptr(),
            rms_final_weight: w.rms_final_weight.raw(),
            // None = the classifier is tied to the embedding table (llama2.rs:203-207)
            output_weight: w.output_weight.as_ref().map_or(ptr::null(), |t| t.raw()),
        };
        let c_conf = ffi::crabml_hip_llama_config_t {
            embedding_dim: conf.embedding_dim,
            hidden_dim: conf.hidden_dim,
            n_layers: conf.n_layers,
            n_heads: conf.n_heads,
            n_kv_heads: conf.n_kv_heads,
            vocab_size: conf.vocab_size,
            seq_len,
            rope_dim: conf.rope_dim.unwrap_or(conf.head_size()),
            rms_norm_eps: conf.rms_norm_eps,
            use_f16_kv_cache: use_f16_kv_cache as i32,
            flags: 0,
            tp_size: 1,
            tp_rank: 0,
            tp_comm: ptr::null_mut(),
            attn_long_from: 0,
            prefill_chunk: 0,
        };
        let mut raw = ptr::null_mut();
        device.check(unsafe {
            ffi::crabml_hip_llama_create(device.raw, &c_conf, &c_weights, &mut raw)
        })?;
        Ok(Self {
            raw,
            conf: conf.clone(),
            device,
            _weights: weights,
            logits: vec![0.0; conf.vocab_size],
        })
    }

    pub fn conf(&self) -> &LlamaConfig {
        &self.conf
    }

    /// llama2.rs:106-108
    pub fn kv_cache_len(&self) -> usize {
        unsafe { ffi::crabml_hip_llama_kv_len(self.raw) }
    }

    /// `Llama2Runner::forward(&[token], pos)` (llama2.rs:184-211): returns the logits of this position
    pub fn forward(&mut self, token: usize, pos: usize) -> Result<&mut [f32]> {
        self.device.check(unsafe {
            ffi::crabml_hip_llama_forward(self.raw, token, pos, self.logits.as_mut_ptr())
        })?;
        Ok(&mut self.logits)
    }

    /// The token loop of `Llama2Runner::prefill` (llama2.rs:124-129) as batched passes: same KV-cache contents, the
    /// logits of the last prompt token.  The tokenizer and the sampler stay with the caller.
    pub fn prefill_tokens(&mut self, tokens: &[usize]) -> Result<&mut [f32]> {
        if tokens.is_empty() {
            bail!(ErrorKind::BadInput, "expected at least 1 prompt token"); // llama2.rs:117-122
        }
        let ids = tokens.iter().map(|t| *t as u32).collect::<Vec<_>>();
        self.device.check(unsafe {
            ffi::crabml_hip_llama_prefill(self.raw, ids.as_ptr(), ids.len(), self.logits.as_mut_ptr())
        })?;
        Ok(&mut self.logits)
    }

    /// The generate loop of llama2.rs:131-182 with temperature 0 (`Llama2Sampler` falls back to the argmax that keeps
    /// the LAST maximum, sampler.rs:109-116): `steps` tokens starting from `token`, sampled on the device, one host
    /// synchronisation at the end.
    pub fn decode_greedy(&mut self, token: usize, steps: usize) -> Result<Vec<usize>> {
        let mut out = vec![0u32; steps];
        self.device.check(unsafe {
            ffi::crabml_hip_llama_decode_greedy(self.raw, token, steps, out.as_mut_ptr())
        })?;
        Ok(out.into_iter().map(|t| t as usize).collect())
    }

    /// empties the KV caches (a new conversation)
    pub fn reset(&mut self) -> Result<()> {
        self.device
            .check(unsafe { ffi::crabml_hip_llama_reset(self.raw) })
    }
}

impl Drop for HipLlamaRunner {
    fn drop(&mut self) {
        unsafe { ffi::crabml_hip_llama_destroy(self.raw) };
    }
}

#[allow(dead_code)]
fn _assert_tensor_impl() {
    fn needs_tensor<T: Tensor>() {}
    needs_tensor::<HipTensor>();
}

//! Raw bindings of include/crabml_hip.h (ABI version 2; the parity / measurement hooks of crabml_hip_debug.h are test
//! infrastructure of the backend repository and are not bound), written by hand: the header is small, plain C
//! (opaque handles, pointers, sizes, fixed-width integers), and a checked-in binding keeps `bindgen` / libclang out
//! of the build.  tests/test_rust_crate.py (backend repository) parses this `extern "C"` block and the header and
//! fails when a name, an arity or an integer width differs.
//!
//! Status codes are `crabml::error::ErrorKind` discriminants (crabml-core/src/error.rs:5-33); GGML type ids are the
//! `#[repr(u32)]` values of `crabml::gguf::GGMLType` (crabml-core/src/gguf.rs:86-108).
#![allow(non_camel_case_types)]
#![allow(dead_code)]

use std::os::raw::c_char;
use std::os::raw::c_void;

pub const CRABML_HIP_ABI_VERSION: i32 = 2;

pub const CRABML_HIP_FLAG_STRICT_ORDER: i32 = 1;
/// every Tensor call launches immediately instead of being recorded (ABI version 1 behaviour)
pub const CRABML_HIP_FLAG_PER_OP: i32 = 2;

pub const CRABML_HIP_LLAMA_NO_GRAPH: i32 = 1;
pub const CRABML_HIP_LLAMA_NO_PREFETCH: i32 = 2;
pub const CRABML_HIP_LLAMA_TP_GRAPH: i32 = 8;
/// fast mode at long context: keep the reference's f16-accumulated PV chain instead of the f32 split-KV kernels
pub const CRABML_HIP_LLAMA_EXACT_ATTENTION: i32 = 4194304;
/// fast mode: RMSNorm's division stays in the producing launch (no deferred 1 / rms)
pub const CRABML_HIP_LLAMA_EXACT_NORM: i32 = 8388608;
/// tensor parallelism: `output_weight` is the rank's vocabulary shard of the classifier (P2P group)
pub const CRABML_HIP_LLAMA_TP_SPLIT_VOCAB: i32 = 1048576;

#[repr(C)]
pub struct crabml_hip_device_t {
    _private: [u8; 0],
}
#[repr(C)]
pub struct crabml_hip_buf_t {
    _private: [u8; 0],
}
#[repr(C)]
pub struct crabml_hip_llama_t {
    _private: [u8; 0],
}
#[repr(C)]
pub struct crabml_hip_tp_comm_t {
    _private: [u8; 0],
}

#[repr(C)]
pub struct crabml_hip_device_options_t {
    pub device_ordinal: i32,
    pub stream: *mut c_void,
    pub flags: i32,
}

/// crabml-llama2/src/model.rs:30-53
#[repr(C)]
pub struct crabml_hip_llama_config_t {
    pub embedding_dim: usize,
    pub hidden_dim: usize,
    pub n_layers: usize,
    pub n_heads: usize,
    pub n_kv_heads: usize,
    pub vocab_size: usize,
    pub seq_len: usize,
    pub rope_dim: usize,
    pub rms_norm_eps: f32,
    pub use_f16_kv_cache: i32,
    pub flags: i32,
    pub tp_size: i32,
    pub tp_rank: i32,
    pub tp_comm: *mut c_void,
    pub attn_long_from: usize,
    pub prefill_chunk: usize,
}

/// crabml-llama2/src/model.rs:55-84; per-layer arrays of n_layers handles
#[repr(C)]
pub struct crabml_hip_llama_weights_t {
    pub token_embed: *const crabml_hip_buf_t,
    pub rms_att_weight: *const *const crabml_hip_buf_t,
    pub rms_ffn_weight: *const *const crabml_hip_buf_t,
    pub wq: *const *const crabml_hip_buf_t,
    pub wk: *const *const crabml_hip_buf_t,
    pub wv: *const *const crabml_hip_buf_t,
    pub wo: *const *const crabml_hip_buf_t,
    pub ffn_gate_weight: *const *const crabml_hip_buf_t,
    pub ffn_down_weight: *const *const crabml_hip_buf_t,
    pub ffn_up_weight: *const *const crabml_hip_buf_t,
    pub rms_final_weight: *const crabml_hip_buf_t,
    pub output_weight: *const crabml_hip_buf_t,
}

extern "C" {
    // ---- device
    pub fn crabml_hip_abi_version() -> i32;
    pub fn crabml_hip_device_create(opts: *const crabml_hip_device_options_t, out: *mut *mut crabml_hip_device_t) -> i32;
    pub fn crabml_hip_device_destroy(dev: *mut crabml_hip_device_t) -> i32;
    pub fn crabml_hip_device_sync(dev: *mut crabml_hip_device_t) -> i32;
    pub fn crabml_hip_last_error(dev: *mut crabml_hip_device_t, buf: *mut c_char, cap: usize) -> usize;
    pub fn crabml_hip_device_stream(dev: *mut crabml_hip_device_t) -> *mut c_void;
    pub fn crabml_hip_device_mem_in_use(dev: *mut crabml_hip_device_t) -> usize;

    // ---- buffers
    pub fn crabml_hip_buf_from_cpu(dev: *mut crabml_hip_device_t, bytes: *const c_void, nbytes: usize, shape: *const usize, ndim: i32, ggml_type: u32, out: *mut *mut crabml_hip_buf_t) -> i32;
    pub fn crabml_hip_buf_alloc(dev: *mut crabml_hip_device_t, n_elems: usize, ggml_type: u32, out: *mut *mut crabml_hip_buf_t) -> i32;
    pub fn crabml_hip_buf_retain(buf: *mut crabml_hip_buf_t) -> i32;
    pub fn crabml_hip_buf_release(buf: *mut crabml_hip_buf_t) -> i32;
    pub fn crabml_hip_buf_dtype(buf: *const crabml_hip_buf_t) -> u32;
    pub fn crabml_hip_buf_len(buf: *const crabml_hip_buf_t) -> usize;

    // ---- data movement
    pub fn crabml_hip_export(dev: *mut crabml_hip_device_t, buf: *const crabml_hip_buf_t, dst: *mut f32, n: usize) -> i32;
    pub fn crabml_hip_export_raw(dev: *mut crabml_hip_device_t, buf: *const crabml_hip_buf_t, dst: *mut c_void, nbytes: usize) -> i32;
    pub fn crabml_hip_dup(dev: *mut crabml_hip_device_t, src: *const crabml_hip_buf_t, out: *mut *mut crabml_hip_buf_t) -> i32;
    pub fn crabml_hip_contiguous(dev: *mut crabml_hip_device_t, src: *const crabml_hip_buf_t, shape: *const usize, strides: *const usize, ndim: i32, out: *mut *mut crabml_hip_buf_t) -> i32;
    pub fn crabml_hip_concatenate(dev: *mut crabml_hip_device_t, dst: *mut crabml_hip_buf_t, dst_shape: *const usize, dst_strides: *const usize, rhs: *const crabml_hip_buf_t, rhs_shape: *const usize, rhs_strides: *const usize, ndim: i32, axis: i32) -> i32;
    pub fn crabml_hip_copy_rows_from(dev: *mut crabml_hip_device_t, dst: *mut crabml_hip_buf_t, src: *const crabml_hip_buf_t, cols: usize, rows: *const usize, n_rows: usize) -> i32;

    // ---- compute
    pub fn crabml_hip_rope_inplace(dev: *mut crabml_hip_device_t, x: *mut crabml_hip_buf_t, n_batch: usize, bi_stride: usize, head_dim: usize, mode: u32, pos: usize, rope_dims: usize) -> i32;
    pub fn crabml_hip_rms_norm_inplace(dev: *mut crabml_hip_device_t, x: *mut crabml_hip_buf_t, rows: usize, cols: usize, eps: f32) -> i32;
    pub fn crabml_hip_softmax_inplace(dev: *mut crabml_hip_device_t, x: *mut crabml_hip_buf_t, rows: usize, cols: usize) -> i32;
    pub fn crabml_hip_silu_inplace(dev: *mut crabml_hip_device_t, x: *mut crabml_hip_buf_t, n: usize) -> i32;
    pub fn crabml_hip_gelu_inplace(dev: *mut crabml_hip_device_t, x: *mut crabml_hip_buf_t, n: usize) -> i32;
    pub fn crabml_hip_mul_inplace(dev: *mut crabml_hip_device_t, a: *mut crabml_hip_buf_t, na: usize, b: *const crabml_hip_buf_t, nb: usize) -> i32;
    pub fn crabml_hip_add_inplace(dev: *mut crabml_hip_device_t, a: *mut crabml_hip_buf_t, na: usize, b: *const crabml_hip_buf_t, nb: usize) -> i32;
    pub fn crabml_hip_scale_inplace(dev: *mut crabml_hip_device_t, a: *mut crabml_hip_buf_t, na: usize, f: f32) -> i32;
    pub fn crabml_hip_matmul_vec(dev: *mut crabml_hip_device_t, w: *const crabml_hip_buf_t, m: usize, k: usize, x: *const crabml_hip_buf_t, b: usize, out: *mut *mut crabml_hip_buf_t) -> i32;
    pub fn crabml_hip_batch_matmul(dev: *mut crabml_hip_device_t, a: *const crabml_hip_buf_t, ba: usize, m: usize, k: usize, b: *const crabml_hip_buf_t, bb: usize, n: usize, sb0: usize, sb1: usize, sb2: usize, out: *mut *mut crabml_hip_buf_t) -> i32;

    // ---- fused Llama decode step
    pub fn crabml_hip_llama_create(dev: *mut crabml_hip_device_t, cfg: *const crabml_hip_llama_config_t, w: *const crabml_hip_llama_weights_t, out: *mut *mut crabml_hip_llama_t) -> i32;
    pub fn crabml_hip_llama_destroy(ctx: *mut crabml_hip_llama_t) -> i32;
    pub fn crabml_hip_llama_forward(ctx: *mut crabml_hip_llama_t, token: usize, pos: usize, logits: *mut f32) -> i32;
    pub fn crabml_hip_llama_decode_greedy(ctx: *mut crabml_hip_llama_t, token: usize, n_steps: usize, out_tokens: *mut u32) -> i32;
    pub fn crabml_hip_llama_prefill(ctx: *mut crabml_hip_llama_t, tokens: *const u32, n: usize, logits: *mut f32) -> i32;
    pub fn crabml_hip_llama_kv_len(ctx: *const crabml_hip_llama_t) -> usize;
    pub fn crabml_hip_llama_reset(ctx: *mut crabml_hip_llama_t) -> i32;

    // ---- tensor-parallel group
    pub fn crabml_hip_tp_get_unique_id(id128: *mut c_void) -> i32;
    pub fn crabml_hip_tp_comm_create(dev: *mut crabml_hip_device_t, id128: *const c_void, nranks: i32, rank: i32, out: *mut *mut crabml_hip_tp_comm_t) -> i32;
    pub fn crabml_hip_tp_comm_destroy(comm: *mut crabml_hip_tp_comm_t) -> i32;
    pub fn crabml_hip_tp_all_reduce(comm: *mut crabml_hip_tp_comm_t, buf: *mut crabml_hip_buf_t, n: usize) -> i32;
    pub fn crabml_hip_tp_p2p_create(dev: *mut crabml_hip_device_t, nranks: i32, rank: i32, max_elems: usize, out: *mut *mut crabml_hip_tp_comm_t) -> i32;
    pub fn crabml_hip_tp_p2p_export(comm: *mut crabml_hip_tp_comm_t, handle64: *mut c_void) -> i32;
    pub fn crabml_hip_tp_p2p_connect(comm: *mut crabml_hip_tp_comm_t, handles: *const c_void) -> i32;
    pub fn crabml_hip_tp_p2p_connect_local(comms: *const *mut crabml_hip_tp_comm_t, n: i32) -> i32;
    pub fn crabml_hip_llama_tp_sim_forward(ranks: *const *mut crabml_hip_llama_t, n: i32, token: usize, pos: usize, logits: *mut f32) -> i32;

}

/*
 * crabml_hip.h -- C ABI of libcrabml_hip.so: the MI355X (gfx950) tensor backend for crabml.
 *
 * This is the drop-in boundary.  A `crabml-hip` Rust crate binds exactly these entry points
 * (bindgen / `extern "C"`) and implements `crabml::tensor::Tensor`
 * (crabml-core/src/tensor/api.rs:11-79) on top of them, so `Llama2Runner<T>`
 * (crabml-llama2/src/llama2.rs:26-43) runs unchanged.  See INTEGRATION.md for the stub.
 *
 * Conventions
 *  - Plain C: opaque handles, pointers and sizes only.  No C++/torch types cross the boundary.
 *  - Every function returns an int status: 0 = OK, otherwise a `crabml_hip_status` value that
 *    maps 1:1 onto crabml::error::ErrorKind (crabml-core/src/error.rs:5-33).  Nothing throws.
 *    crabml_hip_last_error() returns the message of the last failure on that device.
 *  - Shape / stride bookkeeping (TensorStrider, crabml-core/src/tensor/strider.rs) stays on the
 *    host side of the boundary: ops take explicit shapes / strides (in ELEMENTS).
 *  - A `crabml_hip_buf_t` is a reference-counted device allocation + its GGML dtype; cloning a
 *    Rust tensor = retain, dropping = release; views share one buf (with_strider is free).
 *  - Nothing blocks the host but crabml_hip_export / crabml_hip_device_sync / debug snapshots (the same
 *    contract as the wgpu backend: crabml-wgpu/src/wgpu_tensor.rs:293-333), and those are also the only
 *    points where the host can observe data.  Since ABI version 2 the Tensor calls below are therefore
 *    RECORDED (validated, queued, a result handle returned at once) and run when such a point is reached --
 *    op by op, or, when the queue holds the op sequence Llama2Runner::forward issues for one token of a
 *    Llama model (crabml-llama2/src/llama2.rs:184-281, 527-638), as the fused decode step over the
 *    runner's own weight and KV-cache buffers: five launches per layer instead of ~31, enqueued segment by
 *    segment while the host is still recording the next layer (crabml_amd/csrc/lazy.hpp).  Results are
 *    those of the per-op launches bit for bit on a strict-order device; argument errors are still
 *    reported by the call that makes them, device errors by the next export / sync.
 *    A device is driven from one host thread at a time.
 *  - GGML type ids are the #[repr(u32)] values of crabml-core/src/gguf.rs:86-108.
 */
#ifndef CRABML_HIP_H
#define CRABML_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define CRABML_HIP_ABI_VERSION 2 /* 2: Tensor calls are recorded and run at export / sync (below); attn_long_from defaults to 96 */

/* crabml-core/src/error.rs:5-33 */
typedef enum crabml_hip_status {
  CRABML_HIP_OK = 0,
  CRABML_HIP_UNEXPECTED = 1, /* HIP runtime failures land here */
  CRABML_HIP_IO_ERROR = 2,
  CRABML_HIP_TENSOR_NOT_FOUND = 3,
  CRABML_HIP_MODEL_ERROR = 4,
  CRABML_HIP_BAD_INPUT = 5,
  CRABML_HIP_FORMAT_ERROR = 6,
  CRABML_HIP_TENSOR_ERROR = 7, /* shape / dtype misuse (bail!(ErrorKind::TensorError, ..)) */
  CRABML_HIP_CHAT_TEMPLATE_NOT_FOUND = 8,
  CRABML_HIP_NOT_IMPLEMENTED = 9
} crabml_hip_status;

/* crabml-core/src/gguf.rs:86-108 */
typedef enum crabml_hip_ggml_type {
  CRABML_HIP_F32 = 0,
  CRABML_HIP_F16 = 1,
  CRABML_HIP_Q4_0 = 2,
  CRABML_HIP_Q4_1 = 3,
  CRABML_HIP_Q5_0 = 6, /* weights only (rhs: Q8_0), crabml-core/src/cpu/buf/buf_q5_0.rs */
  CRABML_HIP_Q5_1 = 7, /* weights only (rhs: Q8_1), crabml-core/src/cpu/buf/buf_q5_1.rs */
  CRABML_HIP_Q8_0 = 8,
  CRABML_HIP_Q8_1 = 9,
  CRABML_HIP_Q2_K = 10, /* weights only (rhs: Q8_K), crabml-core/src/cpu/buf/buf_q2_k.rs */
  CRABML_HIP_Q3_K = 11, /* weights only (rhs: Q8_K), crabml-core/src/cpu/buf/buf_q3_k.rs */
  CRABML_HIP_Q4_K = 12,
  CRABML_HIP_Q5_K = 13, /* weights only (rhs: Q8_K), in the REFERENCE's field order qs | qh | scales | d | dmin
                         * (crabml-core/src/cpu/buf/buf_q5_k.rs:13-21) -- not ggml's */
  CRABML_HIP_Q6_K = 14, /* weights only (rhs: Q8_K), crabml-core/src/cpu/buf/buf_q6_k.rs */
  CRABML_HIP_Q8_K = 15
} crabml_hip_ggml_type;

/* crabml-core/src/tensor/api.rs:5-9 */
typedef enum crabml_hip_rope_mode { CRABML_HIP_ROPE_LLAMA = 0, CRABML_HIP_ROPE_NEOX = 1 } crabml_hip_rope_mode;

typedef struct crabml_hip_device crabml_hip_device_t;
typedef struct crabml_hip_buf crabml_hip_buf_t;

/* replaces WgpuTensorDeviceOptions (crabml-wgpu/src/wgpu_device.rs:9-38) */
/* flags: CRABML_HIP_FLAG_STRICT_ORDER makes every sum run in the reference's scalar order (matmul_vec adds the
 * per-block terms in block order as vec_dot_*_fallback does; RMSNorm scans and adds its chunks in order;
 * softmax sums sequentially; attention keeps the f16 chains): results BIT-IDENTICAL to the reference's
 * default build end to end (the truncating activation quantizer amplifies 1-ulp re-association differences,
 * see DESIGN.md 2.2).  For Q4_0 / Q8_0 / Q4_1 layers the decode step keeps its five launches per layer
 * (block terms parked in LDS, one lane per row adds them in order): 677 tok/s on the Llama-3-8B shape
 * against 745 for the default.  Default (0) = the fast wave-parallel kernels (re-associated sums). */
#define CRABML_HIP_FLAG_STRICT_ORDER 1
/* every Tensor call launches its kernel(s) immediately instead of being recorded (ABI version 1 behaviour: one launch
 * per call, ~31 per layer; the parity tests of the individual ops and the A/B of the queue use it) */
#define CRABML_HIP_FLAG_PER_OP 2
typedef struct crabml_hip_device_options {
  int32_t device_ordinal; /* HIP device index (one process per GPU: LOCAL_RANK) */
  void* stream;           /* optional caller-owned hipStream_t; NULL = the library creates one */
  int32_t flags;          /* bit set of CRABML_HIP_FLAG_* */
} crabml_hip_device_options_t;

/* ---- device ------------------------------------------------------------------------------ */
int crabml_hip_abi_version(void);
/* replaces WgpuTensorDevice::new (crabml-wgpu/src/wgpu_device.rs:52-75).  Uploads the 65536-entry
 * f16 exp table the reference builds in CpuTensorDevice::init_exp_cache (cpu_device.rs:108-115). */
int crabml_hip_device_create(const crabml_hip_device_options_t* opts, crabml_hip_device_t** out);
int crabml_hip_device_destroy(crabml_hip_device_t* dev);
int crabml_hip_device_sync(crabml_hip_device_t* dev);
/* copies the last error message (NUL terminated, truncated to cap) and returns its full length */
size_t crabml_hip_last_error(crabml_hip_device_t* dev, char* buf, size_t cap);
/* the hipStream_t all work of this device is ordered on (for HIP-event timing / interop) */
void* crabml_hip_device_stream(crabml_hip_device_t* dev);
/* bytes currently held from the HIP allocator (live + pooled) */
size_t crabml_hip_device_mem_in_use(crabml_hip_device_t* dev);

/* ---- buffers ----------------------------------------------------------------------------- */
/* Tensor::from_cpu (api.rs:14-19): uploads `nbytes` of GGML-layout bytes.  Accepts F32, F16,
 * Q8_0, Q4_0, Q4_1, Q5_0, Q5_1, Q2_K, Q3_K, Q4_K, Q5_K, Q6_K, Q8_K -- every weight format of CpuTensorBuf (buf/api.rs:33-47).
 * Quantized tensors must be 2-D (m, k) (or 1-D) with k a multiple
 * of the block size; they are re-laid-out once at upload into 16-byte-aligned planes (quants /
 * scales) -- the unpacked integers and scales are bit-identical to the GGUF bytes. */
int crabml_hip_buf_from_cpu(crabml_hip_device_t* dev, const void* bytes, size_t nbytes, const size_t* shape,
                            int ndim, uint32_t ggml_type, crabml_hip_buf_t** out);
/* Tensor::alloc (api.rs:21-23): F32 (zero filled) or F16 (contents unspecified), n_elems elements */
int crabml_hip_buf_alloc(crabml_hip_device_t* dev, size_t n_elems, uint32_t ggml_type, crabml_hip_buf_t** out);
int crabml_hip_buf_retain(crabml_hip_buf_t* buf);
int crabml_hip_buf_release(crabml_hip_buf_t* buf);
uint32_t crabml_hip_buf_dtype(const crabml_hip_buf_t* buf);
size_t crabml_hip_buf_len(const crabml_hip_buf_t* buf); /* elements */

/* ---- data movement ----------------------------------------------------------------------- */
/* Tensor::export (api.rs:52): copies the first n f32 elements to host memory; BLOCKS. F32 only. */
int crabml_hip_export(crabml_hip_device_t* dev, const crabml_hip_buf_t* buf, float* dst, size_t n);
/* raw export of an F32/F16 buffer's bytes (tests: bit-exact KV-cache / f16 checks); BLOCKS */
int crabml_hip_export_raw(crabml_hip_device_t* dev, const crabml_hip_buf_t* buf, void* dst, size_t nbytes);
/* Tensor::dup (api.rs:55): fresh F32 buffer with a copy of the WHOLE storage (cpu_tensor.rs:333-337) */
int crabml_hip_dup(crabml_hip_device_t* dev, const crabml_hip_buf_t* src, crabml_hip_buf_t** out);
/* Tensor::contiguous (api.rs:40): gathers a 2-/3-D strided F32/F16 view into a fresh dense buffer
 * (contiguous.rs:6-66) */
int crabml_hip_contiguous(crabml_hip_device_t* dev, const crabml_hip_buf_t* src, const size_t* shape,
                          const size_t* strides, int ndim, crabml_hip_buf_t** out);
/* Tensor::concatenate (api.rs:46): writes rhs (strided view) into dst at element offset
 * dst_shape[axis] * dst_strides[axis] (concatenate.rs:12-204).  F32<-F32, F16<-F16, F16<-F32 (RNE).
 * The caller bumps shape[axis] (strider.resize) afterwards. */
int crabml_hip_concatenate(crabml_hip_device_t* dev, crabml_hip_buf_t* dst, const size_t* dst_shape,
                           const size_t* dst_strides, const crabml_hip_buf_t* rhs, const size_t* rhs_shape,
                           const size_t* rhs_strides, int ndim, int axis);
/* Tensor::copy_rows_from (api.rs:50): dst row i <- src row rows[i] (cols elements each), dequantizing
 * quantized sources exactly as BlockQ*::dequantize (cpu_tensor.rs:306-331, buf/api.rs:262-350).
 * dst is F32 or F16. */
int crabml_hip_copy_rows_from(crabml_hip_device_t* dev, crabml_hip_buf_t* dst, const crabml_hip_buf_t* src,
                              size_t cols, const size_t* rows, size_t n_rows);

/* ---- compute (all in place ops require dense F32) ------------------------------------------ */
/* rope.rs:10-80: x viewed as (n_batch, bi_stride) rows holding heads of head_dim; position pos+batch */
int crabml_hip_rope_inplace(crabml_hip_device_t* dev, crabml_hip_buf_t* x, size_t n_batch, size_t bi_stride,
                            size_t head_dim, uint32_t mode, size_t pos, size_t rope_dims);
/* rms_norm.rs:9-47: x /= sqrt(mean(x^2) + eps) per row; cols % 32 == 0 */
int crabml_hip_rms_norm_inplace(crabml_hip_device_t* dev, crabml_hip_buf_t* x, size_t rows, size_t cols, float eps);
/* softmax.rs:11-57 over the last axis; exp through the f16 table */
int crabml_hip_softmax_inplace(crabml_hip_device_t* dev, crabml_hip_buf_t* x, size_t rows, size_t cols);
/* silu.rs:6-13 / gelu.rs:11-17 on the first n elements */
int crabml_hip_silu_inplace(crabml_hip_device_t* dev, crabml_hip_buf_t* x, size_t n);
int crabml_hip_gelu_inplace(crabml_hip_device_t* dev, crabml_hip_buf_t* x, size_t n);
/* arithmetic.rs:5-68: a[i] op= b[i % nb]  (nb == 1: scalar broadcast) */
int crabml_hip_mul_inplace(crabml_hip_device_t* dev, crabml_hip_buf_t* a, size_t na, const crabml_hip_buf_t* b,
                           size_t nb);
int crabml_hip_add_inplace(crabml_hip_device_t* dev, crabml_hip_buf_t* a, size_t na, const crabml_hip_buf_t* b,
                           size_t nb);
/* cpu_tensor.rs:404-410 */
int crabml_hip_scale_inplace(crabml_hip_device_t* dev, crabml_hip_buf_t* a, size_t na, float f);

/* Tensor::matmul_vec (api.rs:76; cpu_tensor.rs:371-386; matmul_vec.rs:9-78):
 * W (m,k) of any supported dtype  x  X (b,k) dense F32  ->  fresh F32 (b,m).
 * X is quantized on the fly to W's vec_dot_rhs_dtype (buf/api.rs:142-159) with the reference's
 * own rounding (Q8_0/Q8_1 truncate, Q8_K rounds half away). */
int crabml_hip_matmul_vec(crabml_hip_device_t* dev, const crabml_hip_buf_t* w, size_t m, size_t k,
                          const crabml_hip_buf_t* x, size_t b, crabml_hip_buf_t** out);
/* Tensor::batch_matmul (api.rs:78; batch_matmul.rs:15-131): A (ba,m,k) dense F32  x  B (bb,k,n)
 * strided F32|F16 (stride_k == 1 or stride_n == 1)  ->  fresh F32 (ba,m,n).
 * Batch broadcast as the reference: F32 B -> bi % bb; F16 B -> bi / (ba/bb). */
int crabml_hip_batch_matmul(crabml_hip_device_t* dev, const crabml_hip_buf_t* a, size_t ba, size_t m, size_t k,
                            const crabml_hip_buf_t* b, size_t bb, size_t n, size_t sb0, size_t sb1, size_t sb2,
                            crabml_hip_buf_t** out);

/* ---- fused Llama decode step (extension of the hot path) ------------------------------------------
 * The trait above costs ~31 launches per layer (one per Tensor call of crabml-llama2/src/llama2.rs:226-271),
 * which makes batch-1 decode launch-bound on MI355X (profiles/r01_trait_path_kernel_trace.md).  This entry
 * point serves the SAME op sequence -- Llama2Runner::forward for the Llama architecture, n_batch = 1
 * (llama2.rs:184-281, 527-638) -- as FIVE kernels per layer replayed from one hipGraph:
 *   q/k/v GEMV + RoPE + scale + KV append | attention (QK^T, softmax, PV, quantize for wo) |
 *   wo GEMV + residual + the ffn RMSNorm + quantize | gate/up GEMV + SiLU*mul + quantize |
 *   ffn_down GEMV + residual + the next layer's RMSNorm + quantize
 * (the norms run in the producing GEMV's epilogue through one in-launch gather; attention switches to three
 * multi-workgroup kernels from `attn_long_from` cached positions), with token id
 * and position living in device memory (greedy argmax on device, sampler.rs:109-116).
 * Arithmetic is the reference's (same rounding points; RoPE cos/sin tabulated on the host with the same
 * libm + iterated-theta recurrence); only GEMV block terms (and, fast mode, the RMSNorm chunk sums and softmax
 * row sums past 1024 positions) are summed wave-parallel, exactly like crabml_hip_matmul_vec.  With
 * CRABML_HIP_FLAG_STRICT_ORDER every sum runs in the reference's scalar order and the step is bit-identical
 * to the reference at every context length.
 * Weights: layer matrices in any matmul_vec format (Q4_0, Q8_0, Q4_1 and Q4_K run fused kernels, Q4_K also
 * with attn_v / ffn_down in Q6_K -- llama.cpp's *_K_M mixes; Q5_0, Q5_1, Q2_K, Q3_K, Q5_K, Q6_K, Q8_K, F16, F32 and any other mix sharing
 * one rhs dtype run as per-op segments inside the same graph); the classifier may have another format; norm
 * weights F32.  Tensor types whose rhs dtypes differ inside a layer: CRABML_HIP_NOT_IMPLEMENTED (use the
 * per-op trait path). */
typedef struct crabml_hip_llama crabml_hip_llama_t;
/* config flags (crabml_hip_llama_config_t.flags).  A/B switches between kernel variants and test hooks share the word: they are
 * declared in crabml_hip_debug.h and are not part of the drop-in surface. */
#define CRABML_HIP_LLAMA_NO_GRAPH 1 /* launch the kernels eagerly instead of replaying a hipGraph */
#define CRABML_HIP_LLAMA_NO_PREFETCH 2 /* do not warm the Infinity Cache from the latency-bound stages */
#define CRABML_HIP_LLAMA_TP_GRAPH 8 /* tp_size > 1: capture the RCCL all-reduces into the hipGraph as well
                                       (default for tp: eager launches; falls back to eager if capture fails) */
#define CRABML_HIP_LLAMA_EXACT_ATTENTION 4194304 /* fast mode, from attn_long_from cached positions: keep the reference's serial
                                          f16-accumulated PV chain (buf_f16.rs:152-163) instead of the split-KV kernels that
                                          accumulate the same f16 products in f32 (DESIGN.md 2.2 states the deviation and its
                                          measured size).  Strict-order devices always run the exact chain. */
#define CRABML_HIP_LLAMA_EXACT_NORM 8388608 /* fast mode, Q4_0 / Q8_0 layers: keep RMSNorm's division inside the launch that produces the row
                                          (one in-launch gather of the chunk sums per wo launch) instead of handing 1 / rms to the consuming
                                          launch -- which quantizes x * w per block first and scales the block scale afterwards: the same
                                          levels up to the 126-vs-127 rounding of a block's largest element.  Strict-order
                                          devices always keep the exact form. */
#define CRABML_HIP_LLAMA_TP_SPLIT_VOCAB 1048576 /* tp_size > 1 (P2P group, or the single-device simulation): the classifier is split by
                                          vocabulary -- weights.output_weight holds rows [tp_rank V / tp_size,
                                          (tp_rank + 1) V / tp_size) of output.weight (V % tp_size == 0, not tied to token_embed);
                                          every rank streams 1 / tp_size of the classifier, takes the arg-max of its shard (last
                                          maximum, sampler.rs:109-116) and the ranks exchange 8-byte {max, index} pairs through the
                                          inboxes (rank order; the later index wins a tie across shards as it does inside one).
                                          Logits copied out by forward(): this rank's shard at its global offsets, -inf elsewhere
                                          (an element-wise max over the ranks' buffers is the all-gather) */
typedef struct crabml_hip_llama_config { /* crabml-llama2/src/model.rs:30-53 */
  size_t embedding_dim, hidden_dim, n_layers, n_heads, n_kv_heads, vocab_size;
  size_t seq_len;  /* KV cache capacity (Llama2Runner::new seq_len, llama2.rs:46-86) */
  size_t rope_dim; /* conf.rope_dim.unwrap_or(head_dim) */
  float rms_norm_eps;
  int32_t use_f16_kv_cache;
  int32_t flags;
  /* tensor parallelism: this context holds rank tp_rank of tp_size shards -- wq/wk/wv split by
   * heads and ffn_gate/ffn_up by rows (column-parallel), wo/ffn_down split along k (row-parallel); the two
   * dim-sized partial sums per layer are all-reduced over tp_comm (RCCL).  tp_size <= 1: single GPU.
   * The weight buffers passed to crabml_hip_llama_create are the LOCAL shards. */
  int32_t tp_size, tp_rank;
  void* tp_comm; /* crabml_hip_tp_comm_t*; NULL with tp_size > 1 = a rank of the single-device simulation */
  size_t attn_long_from; /* cached positions from which attention runs as the multi-workgroup kernels (0 = default: 96 for the
                          * fast step's split-KV kernels, 224 for the exact ones) */
  size_t prefill_chunk;  /* rows per batched prefill pass (0 = default: 1024, 512 on a strict-order device; never more than seq_len) */
} crabml_hip_llama_config_t;
typedef struct crabml_hip_llama_weights { /* crabml-llama2/src/model.rs:55-84; per-layer arrays of n_layers */
  const crabml_hip_buf_t* token_embed;
  const crabml_hip_buf_t* const* rms_att_weight;
  const crabml_hip_buf_t* const* rms_ffn_weight;
  const crabml_hip_buf_t* const* wq;
  const crabml_hip_buf_t* const* wk;
  const crabml_hip_buf_t* const* wv;
  const crabml_hip_buf_t* const* wo;
  const crabml_hip_buf_t* const* ffn_gate_weight;
  const crabml_hip_buf_t* const* ffn_down_weight;
  const crabml_hip_buf_t* const* ffn_up_weight;
  const crabml_hip_buf_t* rms_final_weight;
  const crabml_hip_buf_t* output_weight; /* NULL = tied to token_embed (llama2.rs:203-207) */
} crabml_hip_llama_weights_t;
int crabml_hip_llama_create(crabml_hip_device_t* dev, const crabml_hip_llama_config_t* cfg,
                            const crabml_hip_llama_weights_t* w, crabml_hip_llama_t** out);
int crabml_hip_llama_destroy(crabml_hip_llama_t* ctx);
/* Llama2Runner::forward(&[token], pos): pos must equal the current KV length.  If logits != NULL the
 * vocab_size f32 logits are copied out (BLOCKS); otherwise the call only enqueues. */
int crabml_hip_llama_forward(crabml_hip_llama_t* ctx, size_t token, size_t pos, float* logits);
/* n_steps greedy decode steps on device: forward(token), token = argmax (last maximum), ...; the n_steps
 * sampled ids are written to out_tokens (BLOCKS once, at the end). */
int crabml_hip_llama_decode_greedy(crabml_hip_llama_t* ctx, size_t token, size_t n_steps, uint32_t* out_tokens);
/* The token loop of Llama2Runner::prefill (llama2.rs:111-129: `for (pos, token) in prompt_tokens: forward(&[token],
 * base_pos + pos)`) as batched passes of up to prefill_chunk rows: same KV cache contents and, in logits (nullable,
 * BLOCKS), the logits of the last prompt token.  The reference's own `_batched` flag is ignored (llama2.rs:114)
 * and its batch_matmul has no causal mask for n_batch > 1; here every row attends to the cache up to its own
 * position, which is what the token loop computes.  Weight matrices see the prompt as a (rows, k) rhs: the
 * matmul_vec contract (matmul_vec.rs:6-8), on the matrix cores for Q4_0 / Q8_0 weights and >= 16 rows.  Strict-order
 * devices: bit-identical to the token loop.  tp_size > 1: falls back to the token loop. */
int crabml_hip_llama_prefill(crabml_hip_llama_t* ctx, const uint32_t* tokens, size_t n, float* logits);
size_t crabml_hip_llama_kv_len(const crabml_hip_llama_t* ctx);
int crabml_hip_llama_reset(crabml_hip_llama_t* ctx); /* empties the KV caches */
/* ---- tensor-parallel group over RCCL / xGMI (one process per GPU) ----
 * crabml_hip_tp_get_unique_id fills a 128-byte ncclUniqueId on rank 0; the host broadcasts it (any side channel)
 * and every rank calls crabml_hip_tp_comm_create.  librccl is bound with dlopen on first use. */
typedef struct crabml_hip_tp_comm crabml_hip_tp_comm_t;
int crabml_hip_tp_get_unique_id(void* id128);
int crabml_hip_tp_comm_create(crabml_hip_device_t* dev, const void* id128, int nranks, int rank, crabml_hip_tp_comm_t** out);
int crabml_hip_tp_comm_destroy(crabml_hip_tp_comm_t* comm);
int crabml_hip_tp_all_reduce(crabml_hip_tp_comm_t* comm, crabml_hip_buf_t* buf, size_t n); /* in-place f32 sum */
/* ---- the production collective: one-shot all-reduce over peer-mapped inboxes (no RCCL call on the data path) ----
 * A decode step all-reduces 2 x dim f32 per layer: pure latency.  Every rank allocates an INBOX; its peers map it
 * (hipIpcGetMemHandle / hipIpcOpenMemHandle: a peer GPU's HBM over xGMI, or another process's buffer on the same GPU) and
 * write their partial rows straight into it as 8-byte {f32, epoch} granules; the reader polls its own inbox and adds the
 * rows in rank order, so every rank computes the same bits.  With such a group in crabml_hip_llama_config_t.tp_comm the
 * fast step runs the collective INSIDE the wo / ffn_down kernels (scatter -> gather -> residual -> RMSNorm -> quantize in
 * their epilogue: 5 launches per layer, like one GPU); the per-op segment path (strict order, K-quants) runs it as one small
 * launch where the RCCL group runs ncclAllReduce.  Usage: create on every rank, export the 64-byte handle, ship the handles
 * to every rank (any side channel), connect.  max_elems >= embedding_dim. */
int crabml_hip_tp_p2p_create(crabml_hip_device_t* dev, int nranks, int rank, size_t max_elems, crabml_hip_tp_comm_t** out);
int crabml_hip_tp_p2p_export(crabml_hip_tp_comm_t* comm, void* handle64);
int crabml_hip_tp_p2p_connect(crabml_hip_tp_comm_t* comm, const void* handles /* nranks x 64 bytes, rank order */);
/* ranks living in ONE process (one device object / stream per rank): wires the inboxes as plain device pointers */
int crabml_hip_tp_p2p_connect_local(crabml_hip_tp_comm_t* const* comms, int n);
/* single-device simulation of a tp group (ranks created on ONE device with tp_comm = NULL): same kernels, same
 * sharding, the all-reduce replaced by a local sum -- validates everything but the RCCL transport itself */
int crabml_hip_llama_tp_sim_forward(crabml_hip_llama_t* const* ranks, int n, size_t token, size_t pos, float* logits);

#ifdef __cplusplus
}
#endif
#endif /* CRABML_HIP_H */

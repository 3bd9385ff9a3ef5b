/*
 * crabml_hip_debug.h -- parity hooks, measurement hooks and A/B switches of libcrabml_hip.so.
 *
 * NOT part of the drop-in boundary (include/crabml_hip.h): nothing a crabml maintainer binds lives here.  These entry points
 * and flag bits exist for tests/ (bit-level parity of the production inner loops against the oracle, A/B identity of kernel
 * variants), bench.py (per-kernel event timing, the measured read ceiling) and the labs under tools/.
 */
#ifndef CRABML_HIP_DEBUG_H
#define CRABML_HIP_DEBUG_H

#include "crabml_hip.h"

#ifdef __cplusplus
extern "C" {
#endif

/* ---- device flags of the tests (crabml_hip_device_options_t.flags; the public bits are in crabml_hip.h) ---- */
#define CRABML_HIP_FLAG_LAZY_NO_FUSION 4 /* A/B: record the Tensor calls but always run the queue op by op (never match the decode step) */
#define CRABML_HIP_FLAG_DRY 0x40000000  /* test hook, needs CRABML_HIP_TEST_HOOKS=1: a record-only device object with NO HIP device behind
                                           it -- calls are validated, recorded, matched and counted, nothing is computed, export() yields
                                           zeros.  The CPU test suite drives the queue and the matcher through it. */
/* counters of the recorded-op queue (crabml_amd/csrc/lazy.hpp, LazyStats): out[0..7] = ops recorded, ops run one launch at a time,
 * tokens served by the fused step, recorded ops those tokens replaced, fused segments enqueued, shadow tokens aborted, decode
 * contexts built, final-norm rows bound on demand, [8] nanoseconds the host spent blocked in export, [9] exports served from
 * the pinned logits copy requested at commit, [10] parked decode contexts taken back into service (runners taking turns on one
 * device: up to two contexts wait beside the one being served), [11] contexts dropped because the host had released the model or
 * the caches they served (checked after every flush; also dropped: every idle context when a device allocation fails) */
int crabml_hip_debug_lazy_stats(crabml_hip_device_t* dev, uint64_t* out, size_t cap);

/* the host NUMA node the device is attached to (sysfs numa_node of its PCI function), -1 if unknown: bench.py runs its host
 * threads there (a host that drives the device token by token is sensitive to the socket it sits on) */
int crabml_hip_debug_device_numa_node(crabml_hip_device_t* dev, int32_t* node);

/* ---- parity / debug hooks (used by tests; not on the hot path) ------------------------------ */
/* Quantizes the first n f32 elements of x to `qtype` (Q8_0 | Q8_1 | Q8_K) on the device and returns
 * the blocks in the reference's byte layout (buf_q8_0.rs:8-13, buf_q8_1.rs:73-79, buf_q8_k.rs:6-12). */
int crabml_hip_debug_quantize(crabml_hip_device_t* dev, const crabml_hip_buf_t* x, size_t n, uint32_t qtype,
                              void* dst, size_t dst_bytes);
/* Exact integer part of W(row) . X per 32-element group (one int32 each; k/32 values): the
 * bit-exact gate for the nibble unpack + integer dot. */
int crabml_hip_debug_block_dots(crabml_hip_device_t* dev, const crabml_hip_buf_t* w, size_t m, size_t k, size_t row,
                                const crabml_hip_buf_t* x, int32_t* dst);
/* The integers of the PRODUCTION K-quant loops, per super-block (Q4_K / Q6_K weights, Q8_K rhs; north_star: "bit-exactly
 * at the integer unpack level").  The single-row kernels' own inner loops (rows_partial_q4k / rows_partial_q6k in their
 * debug instantiation: the same code k_gemv_q4_k / k_qkv / k_gemv_res_nq / k_gateup_k_lds run) walk row `row` and hand out
 * what their float part consumes: dst[2 sb] = isum = sum_j scale_j * sum(q * q8) and dst[2 sb + 1] = msum = sum_j min_j *
 * bsum_j for Q4_K (buf_q4_k.rs:212-263; variant 0 = quad-exchanged header dwords, 1 = whole-header loads, the form the
 * LDS-staged kernels use); dst[2 sb] = sum_g scale_g * sum((q6 - 32) * q8), dst[2 sb + 1] = 0 for Q6_K
 * (buf_q6_k.rs:183-234).  k / 256 pairs.  *value (optional) = the kernel's own f32 result for the row. */
int crabml_hip_debug_superblock_ints(crabml_hip_device_t* dev, const crabml_hip_buf_t* w, size_t m, size_t k, size_t row,
                                     const crabml_hip_buf_t* x, int32_t variant, int32_t* dst, float* value);
/* The same integers out of the matrix-core GEMM (k_gemm_mfma_q4k / k_gemm_mfma_q6k themselves, run with their dump
 * pointer set): x holds b >= 16 rows of k; dst[((bi * m + row) * (k / 256) + sb) * 2 + {0, 1}] = (isum, msum) for Q4_K and
 * (sum_g scale_g * sum(q6 * q8), sum_g scale_g * bsum_g) for Q6_K -- the -32 offset is applied as isum - 32 * that.
 * out (optional, b * m floats) receives the GEMM's f32 result. */
int crabml_hip_debug_gemm_ints(crabml_hip_device_t* dev, const crabml_hip_buf_t* w, size_t m, size_t k,
                               const crabml_hip_buf_t* x, size_t b, int32_t* dst, float* out);
/* Sustained HBM read rate of this device as a plain streaming kernel reaches it (16-byte non-temporal loads over `bytes`
 * bytes, best of `reps` launches, HIP events on the device stream): the practical ceiling bench.py quotes next to the
 * 8 TB/s datasheet peak. */
int crabml_hip_debug_read_ceiling(crabml_hip_device_t* dev, size_t bytes, int32_t reps, double* gbytes_per_s);

/* The fast step's long-context attention (k_attn_flash + k_attn_flash_merge, fused_attention.hpp) by itself: one query row per
 * head against `seq` cached positions.  q: n_heads * head_dim f32 (already scaled; rounded to f16 by the kernel as
 * batch_matmul.rs:39 does); k, v: [n_kv][seq][head_dim] f16 bits; slices: the grid's position slices per kv head (1 .. 32; a
 * step uses as many as the context repays); out: n_heads * head_dim f32 = softmax(q k^T) v in f32 arithmetic, from the shipped
 * two-launch form.  out2 (nullable): the same from the single-launch form (the last-arriving workgroup of a kv head merges;
 * launched twice on the same ticket words, which the last arriver re-arms).  Host pointers; blocks. */
int crabml_hip_debug_flash_attention(crabml_hip_device_t* dev, const float* q, const uint16_t* k, const uint16_t* v, size_t n_heads,
                                     size_t n_kv, size_t head_dim, size_t seq, size_t slices, float* out, float* out2);

/* The fast prompt pass's causal attention (k_attn_flash_rows: flash attention on the f16 matrix cores) by itself.  q: rows *
 * n_heads * head_dim f32 (row r is the prompt row at position pos0 + r); k, v: [n_kv][seq_cap][head_dim] f16 bits (the cache
 * after the pass's appends: positions 0 .. pos0 + rows - 1 are live, whatever lies beyond must not matter); out: rows * n_heads *
 * head_dim f32, row r = softmax(q_r k^T over positions 0 .. pos0 + r) v.  head_dim 64 / 128.  Host pointers; blocks. */
int crabml_hip_debug_flash_attention_rows(crabml_hip_device_t* dev, const float* q, const uint16_t* k, const uint16_t* v, size_t n_heads,
                                          size_t n_kv, size_t head_dim, size_t pos0, size_t rows, size_t seq_cap, float* out);

/* ---- A/B switches and test hooks of the fused decode step (crabml_hip_llama_config_t.flags; the public bits are in
 * crabml_hip.h).  Every variant pair is bit-identical unless its comment says otherwise. */
#define CRABML_HIP_LLAMA_NO_NORM_EPILOGUE 4 /* A/B: keep RMSNorm + quantize as its own launch (fast mode runs it in
                                              the wo / ffn_down epilogue; bit-identical either way) */
#define CRABML_HIP_LLAMA_NO_KQUANT_FUSION 256 /* A/B: Q4_K / Q4_1 layers through the per-op segment path */
#define CRABML_HIP_LLAMA_Q4_1_SEGMENTS 512 /* A/B: Q4_1 layers as 11 launches (separate norm / quantize launches) */
#define CRABML_HIP_LLAMA_NO_TILE_ATTENTION 2048 /* A/B: prefill attention as one workgroup per (head, row) */
#define CRABML_HIP_LLAMA_NO_RHS_PROLOGUE 1024 /* A/B: Q4_K layers quantize the rhs of wo / ffn_down in its own launch */
#define CRABML_HIP_LLAMA_TP_DRY_RUN 128 /* measurement hook: a lone tp rank (tp_comm = NULL) steps with its all-reduces
                                          skipped -- per-rank kernel time of a tp group; the logits are meaningless */
#define CRABML_HIP_LLAMA_NO_LONG_ATTENTION 64 /* A/B: one attention workgroup per head at every context length */
#define CRABML_HIP_LLAMA_NO_Q8K_PRODUCERS 32768 /* A/B: Q4_K layers quantize the rhs of wo / ffn_down in those kernels' prologues instead
                                                  of receiving finished Q8_K planes from attention / gate-up (bit-identical) */
#define CRABML_HIP_LLAMA_Q8K_ATTN_PRODUCER 65536 /* A/B, opt-in: the staged attention kernel also assembles wo's Q8_K planes (pairs of heads
                                                   exchange their outputs); measured slower than wo's own 4096-element prologue */
#define CRABML_HIP_LLAMA_NO_PV_PRODUCER_WAVES 131072 /* A/B: long-context decode runs k_attn_pv (the chain wave multiplies and adds)
                                                       instead of k_attn_pv_split (producer waves multiply); bit-identical */
#define CRABML_HIP_LLAMA_NO_PREFILL_ROW_FUSION 262144 /* A/B: the prompt pass keeps residual-add / RMSNorm / quantize and SiLU * mul / quantize as
                                                        separate launches (bit-identical) */
#define CRABML_HIP_LLAMA_NO_PV_ROW_TILES 16384 /* A/B: long-prompt prefill runs the PV pass one prompt row per workgroup */
#define CRABML_HIP_LLAMA_NO_STAGED_ATTENTION 8192 /* A/B: short-context attention without the LDS staging of K / V (k_attn) */
#define CRABML_HIP_LLAMA_FLASH_TICKET 2097152 /* A/B: k_attn_flash merges its partials in the last-arriving workgroup of a kv head
                                                (ticket word, write-through hand-off) instead of a second launch */
#define CRABML_HIP_LLAMA_NO_H_CONSUMER_QUANT 4096 /* A/B, tensor-parallel ranks: gate/up quantizes h itself (hidden / tp / 32 workgroups of 32 rows, k_gateup_q)
                                                   instead of leaving h as f32 from one workgroup per CU for ffn_down's prologue (bit-identical) */
#define CRABML_HIP_LLAMA_PREFILL_INT8_GEMM 524288 /* A/B: the fast prompt pass keeps the bit-exact int8 matrix-core GEMM (with the fused last
                                                     product) at every pass size, instead of the weight-stationary f16 GEMM that Q4_0 / Q8_0 /
                                                     Q4_1 / Q4_K / Q6_K weights take from 32 rows (gemm_f16w.hip; a stated deviation of the
                                                     fast tier) */
#define CRABML_HIP_LLAMA_PREFILL_SEPARATE_F16_ROWS 33554432 /* A/B, fast prompt pass: the rows' f16 planes for the weight GEMM are made by their
                                                              own launch (k_rows_to_f16) instead of by the kernels that quantize the rows,
                                                              and the k pieces of a split wo / ffn_down GEMM are added by their own launch
                                                              (k_addn_f32) instead of by the norm kernel that consumes them (bit-identical) */
#define CRABML_HIP_LLAMA_PREFILL_NO_GU_EPILOGUE 67108864 /* A/B, fast prompt pass: gate | up leave g and u and SiLU * mul (+ quantize) stays its own
                                                           launch, instead of being the f16 GEMM's epilogue (bit-identical) */
#define CRABML_HIP_LLAMA_NO_K_NORM_IN 16777216 /* A/B, fast Q4_K step: wo gathers the row's sums and quantizes its output to Q8_K itself (two
                                                  in-launch hops) instead of leaving x for gate | up to normalize and quantize (bit-identical) */
#define CRABML_HIP_LLAMA_SPLIT_CHUNKS_ALWAYS 16 /* test / tuning hooks for the norm epilogue: two workgroups per */
#define CRABML_HIP_LLAMA_SPLIT_CHUNKS_NEVER 32  /* 32-row chunk always / never (default: only for long rows)   */

/* parity hook: copies the layer's K or V cache (raw f16/f32 bytes, [n_kv_heads][seq_len][head_dim]) */
int crabml_hip_llama_debug_kv(crabml_hip_llama_t* ctx, size_t layer, int32_t which_v, void* dst, size_t nbytes);

/* ---- measurement hook (bench.py `roofline` object) -------------------------------------------------
 * While enabled, every matmul_vec GEMV kernel launch is bracketed by a pair of HIP events recorded on
 * the device's own stream (the stream the kernel runs on); crabml_hip_prof_read() drains them and
 * returns, per weight dtype, the number of launches, the summed kernel time and the summed ALGORITHMIC
 * bytes  m*(k/QK)*BLK + 4k + 4m  (SURVEY.md section 8d).  Costs one event pair per launch: use it in a
 * dedicated instrumented pass, not inside a throughput-timed region. */
typedef struct crabml_hip_prof_entry {
  uint32_t dtype;        /* weight GGML type of the GEMV */
  uint32_t reserved;     /* stage: 0 = matmul_vec; fused step (eager mode only): 1 qkv, 2 wo+res, 3 gate/up, 4 down+res, 5 classifier */
  uint64_t launches;
  double kernel_ms;      /* sum over launches of (stop - start) */
  double algo_bytes;     /* sum over launches of algorithmic bytes */
} crabml_hip_prof_entry_t;
int crabml_hip_prof_enable(crabml_hip_device_t* dev, int on);
/* blocks until the recorded events completed; fills up to cap entries, returns the count in *n */
int crabml_hip_prof_read(crabml_hip_device_t* dev, crabml_hip_prof_entry_t* out, size_t cap, size_t* n);
/* the same drain, one value per launch in record order (for medians / percentiles): fills up to cap durations in
 * milliseconds, returns the count in *n */
int crabml_hip_prof_read_launches(crabml_hip_device_t* dev, float* ms, size_t cap, size_t* n);

#ifdef __cplusplus
}
#endif
#endif /* CRABML_HIP_DEBUG_H */

/*
 * crabml_oracle.h -- CPU ORACLE for the crabml-hip hot path.  TEST INFRASTRUCTURE ONLY.
 *
 * This is a plain-C restatement of the arithmetic of crabml's reference CPU backend
 * (crabml-core/src/cpu/{buf,primitives}) for the block-quantized GEMV path and the ops
 * around it.  It exists so that tests/, __graft_entry__.smoke() and bench.py's
 * `cpu_baseline` leg can check / time-beside the HIP backend.  Nothing under crabml_amd/
 * (the product) may include, link, import or execute anything from this directory.
 *
 * The reference is Rust (nightly-2024-12-31); there is no rustc/cargo in this image, so
 * the reference itself cannot be built (no oracle/_ref).  Every function below cites the
 * reference file:line it follows.  Pinning status:
 *   PINNED by the reference's own known-answer tests (tests/test_oracle_kats.py):
 *     block layouts + dequantize vectors (Q8_0, Q4_0, Q4_1, Q8_1, Q8_K),
 *     quantize_f32_q4_1 bytes, quantize_f32_q8_k (d == 0.0625 + exact round trip),
 *     vec_dot_q8_0_q8_0 == 8978.046 / 3110.453 (exact f32), Q4_K x Q8_K statistical bound,
 *     nearest_i32 x10, get_scale_min_k4, rope / softmax / silu / gelu / rmsnorm / GEMV /
 *     batch_matmul / concatenate / contiguous op goldens.
 *   UNPINNED by any reference test (parity rests on line-by-line restatement only):
 *     vec_dot_q4_0_q8_0 / q4_1_q8_1 / q8_k_q8_k values, quantize_f32_q8_0 / q8_1 outputs,
 *     f16-KV attention numerics, end-to-end strings (15M fixtures are stripped here).
 *
 * Third-party arithmetic restated here: the `half` crate (half = "2.3.1",
 * crabml-core/Cargo.toml:12): f16::from_f32 = IEEE round-to-nearest-even, to_f32 exact,
 * `f16 * f16` / `f16 + f16` = compute in f32, round once to f16.
 */
#ifndef CRABML_ORACLE_H
#define CRABML_ORACLE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* GGML type ids: crabml-core/src/gguf.rs:86-108 */
enum {
  CO_F32 = 0, CO_F16 = 1, CO_Q4_0 = 2, CO_Q4_1 = 3, CO_Q5_0 = 6, CO_Q5_1 = 7, CO_Q8_0 = 8, CO_Q8_1 = 9,
  CO_Q2_K = 10, CO_Q3_K = 11, CO_Q4_K = 12, CO_Q5_K = 13, CO_Q6_K = 14, CO_Q8_K = 15
};

#pragma pack(push, 1)
typedef struct { uint16_t d; int8_t qs[32]; } co_block_q8_0;                 /* buf_q8_0.rs:8-13   34 B */
typedef struct { uint16_t d; uint8_t qs[16]; } co_block_q4_0;                /* buf_q4_0.rs:10-15  18 B */
typedef struct { uint16_t d; uint16_t m; uint8_t qs[16]; } co_block_q4_1;    /* buf_q4_1.rs:10-16  20 B */
typedef struct { uint16_t d; uint16_t s; int8_t qs[32]; } co_block_q8_1;     /* buf_q8_1.rs:73-79  36 B */
typedef struct { uint16_t d; uint16_t dmin; uint8_t scales[12]; uint8_t qs[128]; } co_block_q4_k; /* buf_q4_k.rs:14-21 144 B */
/* the REFERENCE's field order (qs first), which is not ggml's (d, dmin, scales, qh, qs): buf_q5_k.rs:13-21, 176 B */
typedef struct { uint8_t qs[128]; uint8_t qh[32]; uint8_t scales[12]; uint16_t d; uint16_t dmin; } co_block_q5_k;
typedef struct { uint8_t ql[128]; uint8_t qh[64]; int8_t scales[16]; uint16_t d; } co_block_q6_k;       /* buf_q6_k.rs:11-18  210 B */
typedef struct { float d; int8_t qs[256]; int16_t bsums[16]; } co_block_q8_k;                     /* buf_q8_k.rs:6-12  292 B */
typedef struct { uint16_t d; uint8_t qh[4]; uint8_t qs[16]; } co_block_q5_0;                      /* buf_q5_0.rs:13-19  22 B */
typedef struct { uint16_t d; uint16_t m; uint8_t qh[4]; uint8_t qs[16]; } co_block_q5_1;          /* buf_q5_1.rs:10-17  24 B */
typedef struct { uint8_t scales[16]; uint8_t qs[64]; uint16_t d; uint16_t dmin; } co_block_q2_k;  /* buf_q2_k.rs:17-28  84 B */
typedef struct { uint8_t hmask[32]; uint8_t qs[64]; uint8_t scales[12]; uint16_t d; } co_block_q3_k; /* buf_q3_k.rs:19-30 110 B */
#pragma pack(pop)

/* ---- half crate ---- */
uint16_t co_f32_to_f16(float f);
float co_f16_to_f32(uint16_t h);
void co_f32_to_f16_vec(const float* src, uint16_t* dst, size_t n);
void co_f16_to_f32_vec(const uint16_t* src, float* dst, size_t n);

/* ---- block info ---- */
size_t co_block_elems(uint32_t type);   /* 1 for F32/F16 */
size_t co_block_bytes(uint32_t type);
uint32_t co_vec_dot_rhs_dtype(uint32_t type); /* buf/api.rs:142-159 */

/* ---- quantizers (n must be a multiple of the block size) ---- */
void co_quantize_f32_q8_0(const float* x, size_t n, co_block_q8_0* out); /* buf_q8_0.rs:87-134 */
void co_quantize_f32_q8_1(const float* x, size_t n, co_block_q8_1* out); /* buf_q8_1.rs:90-129 */
void co_quantize_f32_q8_k(const float* x, size_t n, co_block_q8_k* out); /* buf_q8_k.rs:84-131 */
void co_quantize_f32_q4_0(const float* x, size_t n, co_block_q4_0* out); /* buf_q4_0.rs:90-124 */
void co_quantize_f32_q4_1(const float* x, size_t n, co_block_q4_1* out); /* buf_q4_1.rs:94-124 */
void co_quantize_f32_q4_k(const float* x, size_t n, co_block_q4_k* out); /* buf_q4_k.rs:111-190 */
void co_quantize_f32_q5_k(const float* x, size_t n, co_block_q5_k* out); /* buf_q5_k.rs:123-227 */
void co_quantize_f32_q5_0(const float* x, size_t n, co_block_q5_0* out); /* buf_q5_0.rs:96-141 */
void co_quantize_f32_q5_1(const float* x, size_t n, co_block_q5_1* out); /* buf_q5_1.rs:99-139 */
/* buf_q2_k.rs:145-214.  Reproduces the reference's indexing slip at :197 (`data[16 * j + ii]` reads the FIRST super-block's
 * values for every super-block): weights quantized by the reference carry it, so the restatement does too. */
void co_quantize_f32_q2_k(const float* x, size_t n, co_block_q2_k* out);
void co_quantize_f32_q3_k(const float* x, size_t n, co_block_q3_k* out); /* buf_q3_k.rs:155-236, util.rs:218-284 */
/* generic: quantize f32 -> `type` into raw bytes; returns 0 ok, -1 unsupported */
int co_quantize(const float* x, size_t n, uint32_t type, void* out);

/* ---- dequantize n elements starting at element `start` (start % block == 0) ---- */
int co_dequantize(const void* blocks, uint32_t type, size_t start, size_t n, float* out);

/* ---- K-quant helpers: buf/util.rs ---- */
int32_t co_nearest_i32(float fval);                                          /* util.rs:10-16 */
void co_get_scale_min_k4(int j, const uint8_t* q, uint8_t* d, uint8_t* m);   /* util.rs:19-27 */

/* ---- dot products, scalar-fallback order (the default build of the reference) ---- */
float co_vec_dot_q8_0_q8_0(const co_block_q8_0* a, const co_block_q8_0* b, size_t nblocks); /* buf_q8_0.rs:275-286 */
float co_vec_dot_q4_0_q8_0(const co_block_q4_0* a, const co_block_q8_0* b, size_t nblocks); /* buf_q4_0.rs:240-253 */
float co_vec_dot_q4_1_q8_1(const co_block_q4_1* a, const co_block_q8_1* b, size_t nblocks); /* buf_q4_1.rs:266-280 */
/* i16_wrap != 0 reproduces the release-build wrap of `bsum * mins as i16` (buf_q4_k.rs:240);
 * *n_overflow (may be NULL) counts products that do not fit i16 (debug builds panic there). */
float co_vec_dot_q4_k_q8_k(const co_block_q4_k* a, const co_block_q8_k* b, size_t nblocks,
                           int i16_wrap, size_t* n_overflow);                                /* buf_q4_k.rs:192-277 */
/* same i16 caveat as Q4_K (buf_q5_k.rs:282-285); the eight f32 lanes aux32 / sums of buf_q5_k.rs:287-318 are kept as written */
float co_vec_dot_q5_k_q8_k(const co_block_q5_k* a, const co_block_q8_k* b, size_t nblocks, int i16_wrap, size_t* n_overflow); /* buf_q5_k.rs:229-325 */
float co_vec_dot_q8_k_q8_k(const co_block_q8_k* a, const co_block_q8_k* b, size_t nblocks); /* buf_q8_k.rs:211-224 */
float co_vec_dot_q5_0_q8_0(const co_block_q5_0* a, const co_block_q8_0* b, size_t nblocks); /* buf_q5_0.rs:143-161 (scalar only) */
float co_vec_dot_q5_1_q8_1(const co_block_q5_1* a, const co_block_q8_1* b, size_t nblocks); /* buf_q5_1.rs:141-160 (scalar only) */
/* `summs` (bsums x the 4-bit mins) is an i16 in the reference (buf_q2_k.rs:219-222): same wrap / overflow reporting as Q4_K */
float co_vec_dot_q2_k_q8_k(const co_block_q2_k* a, const co_block_q8_k* b, size_t nblocks, int i16_wrap, size_t* n_overflow); /* buf_q2_k.rs:216-258 */
float co_vec_dot_q3_k_q8_k(const co_block_q3_k* a, const co_block_q8_k* b, size_t nblocks); /* buf_q3_k.rs:238-329: eight i32 lanes per block, eight f32 sums */
float co_vec_dot_q6_k_q8_k(const co_block_q6_k* a, const co_block_q8_k* b, size_t nblocks); /* buf_q6_k.rs:183-234 (scalar only) */
void co_quantize_f32_q6_k(const float* x, size_t n, co_block_q6_k* out);                    /* buf_q6_k.rs:109-181, util.rs:29-152 */
float co_vec_dot_f32_f32(const float* a, const float* b, size_t n);                          /* buf_f32.rs:19-27 */
float co_vec_dot_f16_f16(const uint16_t* a, const uint16_t* b, size_t n);                    /* buf_f16.rs:83-97 */

/* ---- dot products, x86 AVX2 lane order (RUSTFLAGS=-C target-feature=+avx2 build) ----
 * Used for the CPU baseline and as a second accumulation order in tolerance tests. */
int co_have_avx2(void);
float co_vec_dot_q8_0_q8_0_avx2(const co_block_q8_0* a, const co_block_q8_0* b, size_t nblocks); /* buf_q8_0.rs:228-272 */
float co_vec_dot_q4_0_q8_0_avx2(const co_block_q4_0* a, const co_block_q8_0* b, size_t nblocks); /* buf_q4_0.rs:215-238 */
float co_vec_dot_q8_k_q8_k_avx2(const co_block_q8_k* a, const co_block_q8_k* b, size_t nblocks); /* buf_q8_k.rs:179-209 */

/* ---- exact integer part of the dots: one i32 per 32-element group (bit-exact gate) ----
 * Q4_0: sum_j (nib-8)*q8 ; Q8_0: sum q*q ; Q4_1: sum nib*q8 (unsigned nibbles);
 * Q4_K: per 32-group sum nib*q8 (unscaled, 8 per super-block); Q5_K: the same with the fifth bit (sum q5*q8);
 * Q8_K: per 32-group sum q*q; Q5_0: sum (q5-16)*q8; Q5_1: sum q5*q8;
 * Q2_K / Q3_K: one i32 per 16-element scale group (16 per super-block): sum q2*q8 / sum (q3-4)*q8, unscaled. */
int co_block_dots(const void* w, uint32_t wtype, const void* x, size_t n_elems, int32_t* out);

/* ---- exp / gelu f16 tables: cpu_device.rs:108-125, buf_f32.rs:29-35, gelu.rs:19-22 ---- */
void co_init_exp_cache(uint16_t* table65536);
void co_init_gelu_cache(uint16_t* table65536);
float co_exp_f32_cached(float x, const uint16_t* table);

/* ---- device-like context: thread pool + tables (cpu_device.rs:51-86, thread_pool.rs) ---- */
typedef struct co_device co_device;
co_device* co_device_new(int thread_num, int use_avx2);
void co_device_free(co_device* d);
const uint16_t* co_device_exp_cache(co_device* d);

/* ---- primitives ---- */
/* matmul_vec.rs:9-78: W (m,k) of `wtype` x X (b,k) f32 -> C (b,m) f32.  Quantizes X to the
 * rhs dtype single-threaded on every call, then splits C across thread_num slices. */
int co_matmul_vec(co_device* d, const void* w, uint32_t wtype, size_t m, size_t k,
                  const float* x, size_t b, float* c);
/* batch_matmul.rs:15-131.  A (ba,m,k) contiguous f32; B (bb,k,n) strided, f32 or f16 storage.
 * f32 B: naive loops, B batch index = bi % bb (batch_matmul.rs:61-67).
 * f16 B: A rounded to f16; stride_k==1 -> f32-accumulated dots, B batch = bi/(ba/bb);
 *        stride_n==1 -> f16-accumulated fma (double rounding) (buf_f16.rs:152-163). */
int co_batch_matmul(const float* a, size_t ba, size_t m, size_t k,
                    const void* bdata, uint32_t btype, size_t bb, size_t n,
                    size_t sb0, size_t sb1, size_t sb2, float* c);
void co_rms_norm_inplace(float* x, size_t rows, size_t cols, float eps);     /* rms_norm.rs:9-47 */
/* rope.rs:10-80. mode 0 = Llama, 1 = Neox. x is (n_batch, bi_stride) rows of heads*head_dim */
void co_rope_inplace(float* x, size_t n_batch, size_t bi_stride, size_t head_dim,
                     int mode, size_t pos, size_t rope_dim);
void co_softmax_inplace(co_device* d, float* x, size_t rows, size_t cols);   /* softmax.rs:11-57 */
void co_silu_inplace(co_device* d, float* x, size_t n);                      /* silu.rs:6-13 */
void co_gelu_inplace(co_device* d, float* x, size_t n);                      /* gelu.rs:11-17 */
void co_add_inplace(float* a, size_t na, const float* b, size_t nb);         /* arithmetic.rs:5-35 */
void co_mul_inplace(float* a, size_t na, const float* b, size_t nb);         /* arithmetic.rs:37-68 */
/* concatenate.rs:12-204: writes rhs (strided, f32|f16) into dst (strided, f32|f16) at element
 * offset dshape[axis]*dstrides[axis]; f32->f16 converts RNE.  Returns 0 ok. */
int co_concatenate(void* dst, uint32_t dtype, const size_t* dshape, const size_t* dstrides,
                   const void* rhs, uint32_t rtype, const size_t* rshape, const size_t* rstrides,
                   int ndim, int axis);
/* contiguous.rs:6-66 (elem_size 4 or 2) */
void co_contiguous(const void* src, void* dst, size_t elem_size, const size_t* shape,
                   const size_t* strides, int ndim);

/* greedy argmax, sampler.rs:109-116: Iterator::max_by returns the LAST maximum */
size_t co_argmax_last(const float* x, size_t n);

#ifdef __cplusplus
}
#endif
#endif

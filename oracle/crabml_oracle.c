/*
 * crabml_oracle.c -- CPU ORACLE (test infrastructure only; see crabml_oracle.h).
 *
 * Plain-C restatement of crabml's reference CPU arithmetic.  Build with
 *   gcc -O2 -ffp-contract=off -mavx2 -mfma -mf16c -fPIC -shared   (see oracle/Makefile)
 * -ffp-contract=off matters: Rust never contracts a*b+c into an fma, so neither may we.
 * All citations are into /root/reference (crabml @ 2025-01-03).
 */
#include "crabml_oracle.h"

#include <math.h>
#include <pthread.h>
#include <stdlib.h>
#include <string.h>
#include <immintrin.h>

/* ------------------------------------------------------------------------------------------
 * half crate restatement (half 2.3.1: f16::from_f32 = RNE, NaN keeps sign + payload top bits
 * and gets the quiet bit; to_f32 exact).
 * ---------------------------------------------------------------------------------------- */
uint16_t co_f32_to_f16(float f) {
  uint32_t x;
  memcpy(&x, &f, 4);
  uint32_t sign = (x >> 16) & 0x8000u;
  uint32_t exp = (x >> 23) & 0xffu;
  uint32_t man = x & 0x7fffffu;
  if (exp == 0xffu) {
    if (man == 0) return (uint16_t)(sign | 0x7c00u);
    return (uint16_t)(sign | 0x7e00u | (man >> 13));
  }
  int32_t e = (int32_t)exp - 127 + 15;
  if (e >= 0x1f) return (uint16_t)(sign | 0x7c00u);
  if (e <= 0) {
    if (e < -10) return (uint16_t)sign;
    man |= 0x800000u;
    uint32_t shift = (uint32_t)(14 - e);
    uint32_t half_man = man >> shift;
    uint32_t rem = man & ((1u << shift) - 1u);
    uint32_t halfway = 1u << (shift - 1);
    if (rem > halfway || (rem == halfway && (half_man & 1u))) half_man++;
    return (uint16_t)(sign | half_man);
  }
  uint32_t h = sign | ((uint32_t)e << 10) | (man >> 13);
  uint32_t rem = man & 0x1fffu;
  if (rem > 0x1000u || (rem == 0x1000u && (h & 1u))) h++;
  return (uint16_t)h;
}

float co_f16_to_f32(uint16_t h) {
  uint32_t sign = ((uint32_t)h & 0x8000u) << 16;
  uint32_t exp = (h >> 10) & 0x1fu;
  uint32_t man = h & 0x3ffu;
  uint32_t x;
  if (exp == 0) {
    if (man == 0) {
      x = sign;
    } else {
      int e = -1;
      do {
        e++;
        man <<= 1;
      } while ((man & 0x400u) == 0);
      man &= 0x3ffu;
      x = sign | ((uint32_t)(127 - 15 - e) << 23) | (man << 13);
    }
  } else if (exp == 0x1f) {
    x = sign | 0x7f800000u | (man << 13);
  } else {
    x = sign | ((exp + 127 - 15) << 23) | (man << 13);
  }
  float f;
  memcpy(&f, &x, 4);
  return f;
}

void co_f32_to_f16_vec(const float* src, uint16_t* dst, size_t n) {
  for (size_t i = 0; i < n; i++) dst[i] = co_f32_to_f16(src[i]);
}
void co_f16_to_f32_vec(const uint16_t* src, float* dst, size_t n) {
  for (size_t i = 0; i < n; i++) dst[i] = co_f16_to_f32(src[i]);
}

/* half: `a * b` and `a + b` on f16 = f16::from_f32(a.to_f32() op b.to_f32()) */
static inline uint16_t h_mul(uint16_t a, uint16_t b) { return co_f32_to_f16(co_f16_to_f32(a) * co_f16_to_f32(b)); }
static inline uint16_t h_add(uint16_t a, uint16_t b) { return co_f32_to_f16(co_f16_to_f32(a) + co_f16_to_f32(b)); }

/* ------------------------------------------------------------------------------------------
 * Rust `as` cast semantics
 * ---------------------------------------------------------------------------------------- */
static inline int32_t rs_f32_as_i32(float v) { /* saturating, NaN -> 0 (also std::simd cast) */
  if (v != v) return 0;
  if (v >= 2147483648.0f) return INT32_MAX;
  if (v <= -2147483648.0f) return INT32_MIN;
  return (int32_t)v;
}
static inline int8_t rs_f32_as_i8(float v) {
  if (v != v) return 0;
  if (v >= 127.0f) return 127;
  if (v <= -128.0f) return -128;
  return (int8_t)v;
}
static inline uint8_t rs_f32_as_u8(float v) {
  if (v != v) return 0;
  if (v >= 255.0f) return 255;
  if (v <= 0.0f) return 0;
  return (uint8_t)v;
}
static inline int8_t rs_i32_as_i8(int32_t v) { return (int8_t)(uint8_t)((uint32_t)v & 0xffu); } /* wraps */

/* ------------------------------------------------------------------------------------------ */
size_t co_block_elems(uint32_t t) {
  switch (t) {
    case CO_F32: case CO_F16: return 1;
    case CO_Q4_0: case CO_Q4_1: case CO_Q5_0: case CO_Q5_1: case CO_Q8_0: case CO_Q8_1: return 32;
    case CO_Q2_K: case CO_Q3_K: case CO_Q4_K: case CO_Q5_K: case CO_Q6_K: case CO_Q8_K: return 256;
    default: return 0;
  }
}
size_t co_block_bytes(uint32_t t) {
  switch (t) {
    case CO_F32: return 4;
    case CO_F16: return 2;
    case CO_Q4_0: return sizeof(co_block_q4_0);
    case CO_Q4_1: return sizeof(co_block_q4_1);
    case CO_Q5_0: return sizeof(co_block_q5_0);
    case CO_Q5_1: return sizeof(co_block_q5_1);
    case CO_Q2_K: return sizeof(co_block_q2_k);
    case CO_Q3_K: return sizeof(co_block_q3_k);
    case CO_Q8_0: return sizeof(co_block_q8_0);
    case CO_Q8_1: return sizeof(co_block_q8_1);
    case CO_Q4_K: return sizeof(co_block_q4_k);
    case CO_Q5_K: return sizeof(co_block_q5_k);
    case CO_Q6_K: return sizeof(co_block_q6_k);
    case CO_Q8_K: return sizeof(co_block_q8_k);
    default: return 0;
  }
}
uint32_t co_vec_dot_rhs_dtype(uint32_t t) { /* buf/api.rs:142-159 */
  switch (t) {
    case CO_F32: return CO_F32;
    case CO_F16: return CO_F16;
    case CO_Q8_0: case CO_Q4_0: case CO_Q5_0: return CO_Q8_0;
    case CO_Q8_1: case CO_Q4_1: case CO_Q5_1: return CO_Q8_1;
    case CO_Q8_K: case CO_Q2_K: case CO_Q3_K: case CO_Q4_K: case CO_Q5_K: case CO_Q6_K: return CO_Q8_K;
    default: return 0xffffffffu;
  }
}

/* ------------------------------------------------------------------------------------------
 * Quantizers
 * ---------------------------------------------------------------------------------------- */
/* buf_q8_0.rs:87-134.  max|x| (the simd max tree is order-independent), d = max/127,
 * q = trunc(x / d) via simd cast (saturating, NaN->0) then `as i8` from i32 (wraps). */
void co_quantize_f32_q8_0(const float* x, size_t n, co_block_q8_0* out) {
  for (size_t i = 0; i < n; i += 32) {
    float max = 0.0f;
    for (int j = 0; j < 32; j++) {
      float a = fabsf(x[i + j]);
      max = a > max ? a : max; /* simd_max: NaN-ignoring max; identical for finite data */
    }
    float d = max / 127.0f;
    co_block_q8_0* b = &out[i / 32];
    for (int j = 0; j < 32; j++) {
      float v = x[i + j] / d;
      b->qs[j] = rs_i32_as_i8(rs_f32_as_i32(v));
    }
    b->d = co_f32_to_f16(d);
  }
}

/* buf_q8_1.rs:90-129.  q = clamp(x/d, -128, 127) as i8 (truncation; NaN.max(-128) = -128),
 * s = f16(d * sum(q)) with the sum accumulated in f32 in element order. */
void co_quantize_f32_q8_1(const float* x, size_t n, co_block_q8_1* out) {
  for (size_t i = 0; i < n; i += 32) {
    float max_abs = 0.0f;
    for (int j = 0; j < 32; j++) {
      float a = fabsf(x[i + j]);
      if (a > max_abs) max_abs = a;
    }
    float d = max_abs / 127.0f;
    float s = 0.0f;
    co_block_q8_1* b = &out[i / 32];
    for (int j = 0; j < 32; j++) {
      float sv = x[i + j] / d;
      float c = fminf(fmaxf(sv, -128.0f), 127.0f); /* Rust f32::max/min return the non-NaN operand */
      int8_t q = rs_f32_as_i8(c);
      b->qs[j] = q;
      s += (float)q;
    }
    s *= d;
    b->d = co_f32_to_f16(d);
    b->s = co_f32_to_f16(s);
  }
}

/* buf_q8_k.rs:84-131.  signed value of the max-abs element, scale = -128/max, q = min(round(scale*x),127),
 * round = half away from zero, d = 1/scale; all-zero block -> d = 0, q = 0, bsums = 0. */
void co_quantize_f32_q8_k(const float* x, size_t n, co_block_q8_k* out) {
  for (size_t i = 0; i < n; i += 256) {
    float max_abs = 0.0f, max_value = 0.0f;
    for (int j = 0; j < 256; j++) {
      float a = fabsf(x[i + j]);
      if (a > max_abs) {
        max_abs = a;
        max_value = x[i + j];
      }
    }
    co_block_q8_k* b = &out[i / 256];
    float scale = -128.0f / max_value;
    float d = 1.0f / scale;
    memset(b->qs, 0, sizeof b->qs);
    memset(b->bsums, 0, sizeof b->bsums);
    if (max_abs == 0.0f) {
      d = 0.0f;
    } else {
      for (int j = 0; j < 256; j++) {
        float v = roundf(scale * x[i + j]);
        b->qs[j] = rs_f32_as_i8(fminf(v, 127.0f));
      }
      for (int g = 0; g < 16; g++) {
        int32_t sum = 0;
        for (int j = 0; j < 16; j++) sum += b->qs[g * 16 + j];
        b->bsums[g] = (int16_t)sum;
      }
    }
    b->d = d;
  }
}

/* buf_q4_0.rs:90-124.  d = max|x| / -8 ; id = 1/d ; q = min(15, (x*id + 8.5) as u8) */
void co_quantize_f32_q4_0(const float* x, size_t n, co_block_q4_0* out) {
  for (size_t i = 0; i < n; i += 32) {
    float max_abs = 0.0f;
    for (int j = 0; j < 32; j++) {
      float a = fabsf(x[i + j]);
      if (a > max_abs) max_abs = a;
    }
    float d = max_abs / -8.0f;
    float id = d != 0.0f ? 1.0f / d : 0.0f;
    co_block_q4_0* b = &out[i / 32];
    for (int j = 0; j < 16; j++) {
      float x0 = x[i + j] * id;
      float x1 = x[i + 16 + j] * id;
      uint8_t xi0 = rs_f32_as_u8(x0 + 8.5f);
      uint8_t xi1 = rs_f32_as_u8(x1 + 8.5f);
      if (xi0 > 15) xi0 = 15;
      if (xi1 > 15) xi1 = 15;
      b->qs[j] = (uint8_t)(xi0 | (xi1 << 4));
    }
    b->d = co_f32_to_f16(d);
  }
}

/* buf_q4_1.rs:94-124.  NOTE the reference packs elements (2i, 2i+1) into byte i -- interleaved,
 * unlike its own vec_dot (lo -> j, hi -> j+16).  Restated as written. */
void co_quantize_f32_q4_1(const float* x, size_t n, co_block_q4_1* out) {
  for (size_t i = 0; i < n; i += 32) {
    float mn = 3.40282347e+38f, mx = -3.40282347e+38f; /* f32::MAX, f32::MIN */
    for (int j = 0; j < 32; j++) {
      mn = fminf(x[i + j], mn);
      mx = fmaxf(x[i + j], mx);
    }
    float d = (mx - mn) / 15.0f;
    float id = d != 0.0f ? 1.0f / d : 0.0f;
    co_block_q4_1* b = &out[i / 32];
    for (int j = 0; j < 32; j += 2) {
      uint8_t v0 = rs_f32_as_u8(fminf(roundf((x[i + j] - mn) * id), 15.0f));
      uint8_t v1 = rs_f32_as_u8(fminf(roundf((x[i + j + 1] - mn) * id), 15.0f));
      b->qs[j / 2] = (uint8_t)(v0 | (v1 << 4));
    }
    b->d = co_f32_to_f16(d);
    b->m = co_f32_to_f16(mn);
  }
}

/* util.rs:10-16.  NOTE: the reference does a numeric `as i32` (not ggml's bit reinterpretation);
 * the float add rounds to an integer (RNE) because 2^23 <= fval + 12582912 < 2^24. */
int32_t co_nearest_i32(float fval) {
  int32_t i = rs_f32_as_i32(fval + 12582912.0f);
  return (i & 0x007fffff) - 0x00400000;
}

void co_get_scale_min_k4(int j, const uint8_t* q, uint8_t* d, uint8_t* m) { /* util.rs:19-27 */
  if (j < 4) {
    *d = q[j] & 63;
    *m = q[j + 4] & 63;
  } else {
    *d = (uint8_t)((q[j + 4] & 0xF) | ((q[j - 4] >> 6) << 4));
    *m = (uint8_t)((q[j + 4] >> 4) | ((q[j] >> 6) << 4));
  }
}

/* util.rs:154-216 */
static float make_qkx1_quants(int n, int nmax, const float* data, uint8_t* l, float* the_min, int ntry) {
  float min = data[0], max = data[0];
  for (int i = 0; i < n; i++) {
    if (data[i] < min) min = data[i];
    if (data[i] > max) max = data[i];
  }
  if (max == min) {
    for (int i = 0; i < n; i++) l[i] = 0;
    *the_min = 0.0f;
    return 0.0f;
  }
  if (min > 0.0f) min = 0.0f;
  float iscale = (float)nmax / (max - min);
  float scale = 1.0f / iscale;
  for (int t = 0; t < ntry; t++) {
    float sumlx = 0.0f;
    int32_t suml2 = 0;
    int did_change = 0;
    for (int i = 0; i < n; i++) {
      int32_t li = co_nearest_i32(iscale * (data[i] - min));
      li = li < nmax ? li : nmax;
      li = li > 0 ? li : 0;
      if ((uint8_t)li != l[i]) {
        l[i] = (uint8_t)li;
        did_change = 1;
      }
      sumlx += (data[i] - min) * (float)li;
      suml2 += li * li;
    }
    scale = sumlx / (float)suml2;
    float sum = 0.0f;
    for (int i = 0; i < n; i++) sum += data[i] - scale * (float)l[i];
    min = sum / (float)n;
    if (min > 0.0f) min = 0.0f;
    iscale = 1.0f / scale;
    if (!did_change) break;
  }
  *the_min = -min;
  return scale;
}

/* buf_q4_k.rs:111-190 */
void co_quantize_f32_q4_k(const float* x, size_t n, co_block_q4_k* out) {
  float scales[8], mins[8];
  memset(scales, 0, sizeof scales);
  memset(mins, 0, sizeof mins);
  for (size_t c = 0; c < n; c += 256) {
    const float* chunk = x + c;
    uint8_t l[256];
    memset(l, 0, sizeof l);
    float max_scale = 0.0f, max_min = 0.0f;
    uint8_t bs[12];
    memset(bs, 0, sizeof bs);
    for (int ib = 0; ib < 8; ib++) {
      scales[ib] = make_qkx1_quants(32, 15, chunk + 32 * ib, l + 32 * ib, &mins[ib], 5);
      if (scales[ib] > max_scale) max_scale = scales[ib];
      if (mins[ib] > max_min) max_min = mins[ib];
    }
    float inv_scale = max_scale > 0.0f ? 63.0f / max_scale : 0.0f;
    float inv_min = max_min > 0.0f ? 63.0f / max_min : 0.0f;
    for (int idx = 0; idx < 8; idx++) {
      int32_t a = co_nearest_i32(inv_scale * scales[idx]);
      int32_t b = co_nearest_i32(inv_min * mins[idx]);
      uint8_t ls = (uint8_t)(a < 63 ? a : 63);
      uint8_t lm = (uint8_t)(b < 63 ? b : 63);
      if (idx < 4) {
        bs[idx] = ls;
        bs[idx + 4] = lm;
      } else {
        bs[idx + 4] = (uint8_t)((ls & 0xF) | ((lm & 0xF) << 4));
        bs[idx - 4] |= (uint8_t)((ls >> 4) << 6);
        bs[idx] |= (uint8_t)((lm >> 4) << 6);
      }
    }
    float d = max_scale / 63.0f;
    float dmin = max_min / 63.0f;
    for (int idx = 0; idx < 8; idx++) {
      uint8_t sc, m;
      co_get_scale_min_k4(idx, bs, &sc, &m);
      float dd = d * (float)sc;
      if (dd == 0.0f) continue;
      float dm = dmin * (float)m;
      for (int i = 0; i < 32; i++) {
        int index = 32 * idx + i;
        int32_t ll = co_nearest_i32((chunk[index] + dm) / dd);
        ll = ll < 0 ? 0 : (ll > 15 ? 15 : ll);
        l[index] = (uint8_t)ll;
      }
    }
    co_block_q4_k* b = &out[c / 256];
    for (int q = 0; q < 4; q++)
      for (int id = 0; id < 32; id++) b->qs[32 * q + id] = (uint8_t)(l[64 * q + id] | (l[64 * q + id + 32] << 4));
    b->d = co_f32_to_f16(d);
    b->dmin = co_f32_to_f16(dmin);
    memcpy(b->scales, bs, 12);
  }
}

/* buf_q5_k.rs:123-227: make_qkx1_quants(32, 31, .., 9) per 32 values, 6-bit scales / mins packed as in Q4_K, then the 5-bit
 * levels re-derived from the rounded scales; the low nibbles go to qs (two 32-value halves of a 64-chunk per byte), bit 4 to qh
 * (mask m1 = 1 << 2c for the first half of chunk c, m2 = 2 << 2c for the second). */
void co_quantize_f32_q5_k(const float* x, size_t n, co_block_q5_k* out) {
  float scales[8], mins[8];
  memset(scales, 0, sizeof scales);
  memset(mins, 0, sizeof mins);
  for (size_t c = 0; c < n; c += 256) {
    const float* chunk = x + c;
    uint8_t l[256];
    memset(l, 0, sizeof l);
    float max_scale = 0.0f, max_min = 0.0f;
    uint8_t bs[12];
    memset(bs, 0, sizeof bs);
    for (int ib = 0; ib < 8; ib++) {
      scales[ib] = make_qkx1_quants(32, 31, chunk + 32 * ib, l + 32 * ib, &mins[ib], 9);
      if (scales[ib] > max_scale) max_scale = scales[ib];
      if (mins[ib] > max_min) max_min = mins[ib];
    }
    float inv_scale = max_scale > 0.0f ? 63.0f / max_scale : 0.0f;
    float inv_min = max_min > 0.0f ? 63.0f / max_min : 0.0f;
    for (int idx = 0; idx < 8; idx++) {
      int32_t a = co_nearest_i32(inv_scale * scales[idx]);
      int32_t b = co_nearest_i32(inv_min * mins[idx]);
      uint8_t ls = (uint8_t)(a < 63 ? a : 63);
      uint8_t lm = (uint8_t)(b < 63 ? b : 63);
      if (idx < 4) {
        bs[idx] = ls;
        bs[idx + 4] = lm;
      } else {
        bs[idx + 4] = (uint8_t)((ls & 0xF) | ((lm & 0xF) << 4));
        bs[idx - 4] |= (uint8_t)((ls >> 4) << 6);
        bs[idx] |= (uint8_t)((lm >> 4) << 6);
      }
    }
    float d = max_scale / 63.0f;
    float dmin = max_min / 63.0f;
    for (int idx = 0; idx < 8; idx++) {
      uint8_t sc, m;
      co_get_scale_min_k4(idx, bs, &sc, &m);
      float dd = d * (float)sc;
      if (dd == 0.0f) continue;
      float dm = dmin * (float)m;
      for (int i = 0; i < 32; i++) {
        int index = 32 * idx + i;
        int32_t ll = co_nearest_i32((chunk[index] + dm) / dd);
        ll = ll < 0 ? 0 : (ll > 31 ? 31 : ll);
        l[index] = (uint8_t)ll;
      }
    }
    co_block_q5_k* b = &out[c / 256];
    memset(b->qh, 0, sizeof b->qh);
    uint8_t m1 = 1, m2 = 2;
    for (int q = 0; q < 4; q++) {
      for (int id = 0; id < 32; id++) {
        uint8_t l1 = l[64 * q + id], l2 = l[64 * q + id + 32];
        if (l1 > 15) {
          l1 = (uint8_t)(l1 - 16);
          b->qh[id] |= m1;
        }
        if (l2 > 15) {
          l2 = (uint8_t)(l2 - 16);
          b->qh[id] |= m2;
        }
        b->qs[32 * q + id] = (uint8_t)(l1 | (l2 << 4));
      }
      m1 = (uint8_t)(m1 << 2);
      m2 = (uint8_t)(m2 << 2);
    }
    b->d = co_f32_to_f16(d);
    b->dmin = co_f32_to_f16(dmin);
    memcpy(b->scales, bs, 12);
  }
}

int co_quantize(const float* x, size_t n, uint32_t type, void* out) {
  switch (type) {
    case CO_F32: memcpy(out, x, n * 4); return 0;
    case CO_F16: co_f32_to_f16_vec(x, (uint16_t*)out, n); return 0; /* buf_f16.rs:34-39 */
    case CO_Q8_0: co_quantize_f32_q8_0(x, n, (co_block_q8_0*)out); return 0;
    case CO_Q8_1: co_quantize_f32_q8_1(x, n, (co_block_q8_1*)out); return 0;
    case CO_Q8_K: co_quantize_f32_q8_k(x, n, (co_block_q8_k*)out); return 0;
    case CO_Q4_0: co_quantize_f32_q4_0(x, n, (co_block_q4_0*)out); return 0;
    case CO_Q4_1: co_quantize_f32_q4_1(x, n, (co_block_q4_1*)out); return 0;
    case CO_Q4_K: co_quantize_f32_q4_k(x, n, (co_block_q4_k*)out); return 0;
    case CO_Q5_K: co_quantize_f32_q5_k(x, n, (co_block_q5_k*)out); return 0;
    case CO_Q6_K: co_quantize_f32_q6_k(x, n, (co_block_q6_k*)out); return 0;
    case CO_Q5_0: co_quantize_f32_q5_0(x, n, (co_block_q5_0*)out); return 0;
    case CO_Q5_1: co_quantize_f32_q5_1(x, n, (co_block_q5_1*)out); return 0;
    case CO_Q2_K: co_quantize_f32_q2_k(x, n, (co_block_q2_k*)out); return 0;
    case CO_Q3_K: co_quantize_f32_q3_k(x, n, (co_block_q3_k*)out); return 0;
    default: return -1;
  }
}

static uint32_t rd_u32le(const uint8_t* p) { return (uint32_t)p[0] | ((uint32_t)p[1] << 8) | ((uint32_t)p[2] << 16) | ((uint32_t)p[3] << 24); }
static void dq_q5_0(const co_block_q5_0* b, float* o) { /* buf_q5_0.rs:22-37 */
  const float d = co_f16_to_f32(b->d);
  const uint32_t qh = rd_u32le(b->qh);
  for (int i = 0; i < 16; i++) {
    const uint8_t xh0 = (uint8_t)(((qh >> i) << 4) & 0x10), xh1 = (uint8_t)((qh >> (i + 12)) & 0x10);
    const int32_t x0 = (int32_t)((b->qs[i] & 0x0F) | xh0) - 16, x1 = (int32_t)((b->qs[i] >> 4) | xh1) - 16;
    o[i] = (float)x0 * d;
    o[i + 16] = (float)x1 * d;
  }
}
static void dq_q5_1(const co_block_q5_1* b, float* o) { /* buf_q5_1.rs:20-36 (NOT interleaved, unlike Q4_1's) */
  const float d = co_f16_to_f32(b->d), m = co_f16_to_f32(b->m);
  const uint32_t qh = rd_u32le(b->qh);
  for (int i = 0; i < 16; i++) {
    const uint8_t xh0 = (uint8_t)(((qh >> i) << 4) & 0x10), xh1 = (uint8_t)((qh >> (i + 12)) & 0x10);
    const uint8_t x0 = (uint8_t)((b->qs[i] & 0x0F) | xh0), x1 = (uint8_t)((b->qs[i] >> 4) | xh1);
    o[i] = (float)x0 * d + m;
    o[i + 16] = (float)x1 * d + m;
  }
}
static void dq_q2_k(const co_block_q2_k* b, float* o) { /* buf_q2_k.rs:35-69 */
  const float d = co_f16_to_f32(b->d), mn = co_f16_to_f32(b->dmin);
  int is = 0, oi = 0;
  for (int half = 0; half < 2; half++) {
    const uint8_t* qs = b->qs + 32 * half;
    for (int shift = 0; shift < 8; shift += 2) {
      for (int h = 0; h < 2; h++) {
        const uint8_t sc = b->scales[is++];
        const float dl = d * (float)(sc & 0xF), ml = mn * (float)(sc >> 4);
        for (int l = 0; l < 16; l++) o[oi++] = dl * (float)((qs[16 * h + l] >> shift) & 3) - ml;
      }
    }
  }
}
/* the 16 6-bit scales of a Q3_K block, as the reference's u32 shuffle leaves them (buf_q3_k.rs:44-55 / :289-301) */
static void q3k_scales(const uint8_t* s12, int8_t* sc16) {
  for (int j = 0; j < 16; j++) {
    const uint8_t lo = j < 8 ? (uint8_t)(s12[j] & 0xF) : (uint8_t)(s12[j - 8] >> 4);
    const uint8_t hi = (uint8_t)((s12[8 + j % 4] >> (2 * (j / 4))) & 3);
    sc16[j] = (int8_t)(lo | (hi << 4));
  }
}
static void dq_q3_k(const co_block_q3_k* b, float* o) { /* buf_q3_k.rs:37-88 */
  const float d_all = co_f16_to_f32(b->d);
  int8_t sc[16];
  q3k_scales(b->scales, sc);
  uint8_t m = 1;
  int is = 0, oi = 0;
  for (int half = 0; half < 2; half++) {
    const uint8_t* qs = b->qs + 32 * half;
    for (int shift = 0; shift < 8; shift += 2) {
      for (int h = 0; h < 2; h++) {
        const float dl = d_all * (float)(int8_t)(sc[is++] - 32);
        for (int l = 0; l < 16; l++) {
          const int8_t mm = (b->hmask[16 * h + l] & m) ? 0 : 4;
          o[oi++] = dl * (float)(int8_t)((int8_t)((qs[16 * h + l] >> shift) & 3) - mm);
        }
      }
      m = (uint8_t)(m << 1);
    }
  }
}

/* ------------------------------------------------------------------------------------------
 * Dequantize (per block), then slice.  Each value is ONE rounding of int * f32(d) (or two for
 * the affine formats), exactly as written in the reference.
 * ---------------------------------------------------------------------------------------- */
static void dq_q8_0(const co_block_q8_0* b, float* o) { /* buf_q8_0.rs:18-23 */
  float d = co_f16_to_f32(b->d);
  for (int i = 0; i < 32; i++) o[i] = (float)b->qs[i] * d;
}
static void dq_q4_0(const co_block_q4_0* b, float* o) { /* buf_q4_0.rs:18-27 */
  float d = co_f16_to_f32(b->d);
  for (int i = 0; i < 16; i++) {
    int x0 = (b->qs[i] & 0x0F) - 8;
    int x1 = (b->qs[i] >> 4) - 8;
    o[i] = (float)x0 * d;
    o[i + 16] = (float)x1 * d;
  }
}
static void dq_q4_1(const co_block_q4_1* b, float* o) { /* buf_q4_1.rs:19-30 (interleaved, as written) */
  float d = co_f16_to_f32(b->d), m = co_f16_to_f32(b->m);
  for (int i = 0; i < 16; i++) {
    float x0 = (float)(b->qs[i] & 0x0F);
    float x1 = (float)((b->qs[i] >> 4) & 0x0F);
    o[2 * i] = x0 * d + m;
    o[2 * i + 1] = x1 * d + m;
  }
}
static void dq_q8_1(const co_block_q8_1* b, float* o) { /* buf_q8_1.rs:82-87 */
  float d = co_f16_to_f32(b->d);
  for (int i = 0; i < 32; i++) o[i] = (float)b->qs[i] * d;
}
static void dq_q4_k(const co_block_q4_k* b, float* o) { /* buf_q4_k.rs:24-47 (stray println! not reproduced) */
  float d = co_f16_to_f32(b->d), min = co_f16_to_f32(b->dmin);
  int is = 0;
  for (int c = 0; c < 4; c++) {
    uint8_t sc, m;
    co_get_scale_min_k4(is, b->scales, &sc, &m);
    float d1 = d * (float)sc, m1 = min * (float)m;
    co_get_scale_min_k4(is + 1, b->scales, &sc, &m);
    float d2 = d * (float)sc, m2 = min * (float)m;
    const uint8_t* q = b->qs + 32 * c;
    float* oc = o + 64 * c;
    for (int l = 0; l < 32; l++) {
      oc[l] = d1 * (float)(q[l] & 0xF) - m1;
      oc[l + 32] = d2 * (float)(q[l] >> 4) - m2;
    }
    is += 2;
  }
}
static void dq_q5_k(const co_block_q5_k* b, float* o) { /* buf_q5_k.rs:24-63 (stray println! not reproduced) */
  float d = co_f16_to_f32(b->d), min = co_f16_to_f32(b->dmin);
  int is = 0;
  uint8_t u1 = 1, u2 = 2;
  for (int c = 0; c < 4; c++) {
    uint8_t sc, m;
    co_get_scale_min_k4(is, b->scales, &sc, &m);
    float d1 = d * (float)sc, m1 = min * (float)m;
    co_get_scale_min_k4(is + 1, b->scales, &sc, &m);
    float d2 = d * (float)sc, m2 = min * (float)m;
    const uint8_t* q = b->qs + 32 * c;
    float* oc = o + 64 * c;
    for (int l = 0; l < 32; l++) {
      oc[l] = d1 * ((float)(q[l] & 0xF) + ((b->qh[l] & u1) ? 16.0f : 0.0f)) - m1;
      oc[l + 32] = d2 * ((float)(q[l] >> 4) + ((b->qh[l] & u2) ? 16.0f : 0.0f)) - m2;
    }
    is += 2;
    u1 = (uint8_t)(u1 << 2);
    u2 = (uint8_t)(u2 << 2);
  }
}
static void dq_q8_k(const co_block_q8_k* b, float* o) { /* buf_q8_k.rs:15-20 */
  for (int i = 0; i < 256; i++) o[i] = b->d * (float)b->qs[i];
}

static void dq_q6_k(const co_block_q6_k* b, float* o) { /* buf_q6_k.rs:21-48 */
  const float d = co_f16_to_f32(b->d);
  for (int idx = 0; idx < 2; idx++) {
    float* buf = o + 128 * idx;
    const int8_t* sc = b->scales + 8 * idx;
    const uint8_t* ql = b->ql + 64 * idx;
    const uint8_t* qh = b->qh + 32 * idx;
    for (int l = 0; l < 32; l++) {
      int is = l / 16;
      int8_t q1 = (int8_t)((int8_t)((ql[l] & 0xF) | ((qh[l] & 3) << 4)) - 32);
      int8_t q2 = (int8_t)((int8_t)((ql[l + 32] & 0xF) | (((qh[l] >> 2) & 3) << 4)) - 32);
      int8_t q3 = (int8_t)((int8_t)((ql[l] >> 4) | (((qh[l] >> 4) & 3) << 4)) - 32);
      int8_t q4 = (int8_t)((int8_t)((ql[l + 32] >> 4) | (((qh[l] >> 6) & 3) << 4)) - 32);
      buf[l] = d * (float)sc[is] * (float)q1; /* left to right: (d * scale) * q */
      buf[l + 32] = d * (float)sc[is + 2] * (float)q2;
      buf[l + 64] = d * (float)sc[is + 4] * (float)q3;
      buf[l + 96] = d * (float)sc[is + 6] * (float)q4;
    }
  }
}

int co_dequantize(const void* blocks, uint32_t type, size_t start, size_t n, float* out) {
  size_t be = co_block_elems(type);
  if (be == 0) return -1;
  if (type == CO_F32) {
    memcpy(out, (const float*)blocks + start, n * 4);
    return 0;
  }
  if (type == CO_F16) {
    co_f16_to_f32_vec((const uint16_t*)blocks + start, out, n);
    return 0;
  }
  if (start % be != 0) return -2;
  float tmp[256];
  size_t bi = start / be, done = 0;
  const uint8_t* p = (const uint8_t*)blocks;
  size_t bb = co_block_bytes(type);
  while (done < n) {
    const void* blk = p + bi * bb;
    switch (type) {
      case CO_Q8_0: dq_q8_0((const co_block_q8_0*)blk, tmp); break;
      case CO_Q4_0: dq_q4_0((const co_block_q4_0*)blk, tmp); break;
      case CO_Q4_1: dq_q4_1((const co_block_q4_1*)blk, tmp); break;
      case CO_Q8_1: dq_q8_1((const co_block_q8_1*)blk, tmp); break;
      case CO_Q4_K: dq_q4_k((const co_block_q4_k*)blk, tmp); break;
      case CO_Q5_K: dq_q5_k((const co_block_q5_k*)blk, tmp); break;
      case CO_Q6_K: dq_q6_k((const co_block_q6_k*)blk, tmp); break;
      case CO_Q8_K: dq_q8_k((const co_block_q8_k*)blk, tmp); break;
      case CO_Q5_0: dq_q5_0((const co_block_q5_0*)blk, tmp); break;
      case CO_Q5_1: dq_q5_1((const co_block_q5_1*)blk, tmp); break;
      case CO_Q2_K: dq_q2_k((const co_block_q2_k*)blk, tmp); break;
      case CO_Q3_K: dq_q3_k((const co_block_q3_k*)blk, tmp); break;
      default: return -1;
    }
    size_t take = n - done < be ? n - done : be;
    memcpy(out + done, tmp, take * 4);
    done += take;
    bi++;
  }
  return 0;
}

/* ------------------------------------------------------------------------------------------
 * Dots, scalar-fallback order
 * ---------------------------------------------------------------------------------------- */
float co_vec_dot_q8_0_q8_0(const co_block_q8_0* a, const co_block_q8_0* b, size_t nb) {
  float sumf = 0.0f;
  for (size_t i = 0; i < nb; i++) {
    int32_t sumi = 0;
    for (int j = 0; j < 32; j++) sumi += (int32_t)a[i].qs[j] * (int32_t)b[i].qs[j];
    sumf += (float)sumi * co_f16_to_f32(a[i].d) * co_f16_to_f32(b[i].d);
  }
  return sumf;
}

float co_vec_dot_q4_0_q8_0(const co_block_q4_0* a, const co_block_q8_0* b, size_t nb) {
  float sumf = 0.0f;
  for (size_t i = 0; i < nb; i++) {
    int32_t sumi = 0;
    for (int j = 0; j < 16; j++) {
      int32_t v0 = (int32_t)(a[i].qs[j] & 0x0F) - 8;
      int32_t v1 = (int32_t)(a[i].qs[j] >> 4) - 8;
      sumi += v0 * (int32_t)b[i].qs[j] + v1 * (int32_t)b[i].qs[j + 16];
    }
    sumf += (float)sumi * co_f16_to_f32(a[i].d) * co_f16_to_f32(b[i].d);
  }
  return sumf;
}

/* buf_q4_1.rs:266-280: (d_a * d_b) and (m * s) are f16*f16 products ROUNDED TO f16 (half crate). */
float co_vec_dot_q4_1_q8_1(const co_block_q4_1* a, const co_block_q8_1* b, size_t nb) {
  float sumf = 0.0f;
  for (size_t i = 0; i < nb; i++) {
    int32_t sumi = 0;
    for (int j = 0; j < 16; j++) {
      int32_t v0 = (int32_t)(a[i].qs[j] & 0x0F);
      int32_t v1 = (int32_t)((a[i].qs[j] >> 4) & 0x0F);
      sumi += v0 * (int32_t)b[i].qs[j] + v1 * (int32_t)b[i].qs[j + 16];
    }
    sumf += co_f16_to_f32(h_mul(a[i].d, b[i].d)) * (float)sumi + co_f16_to_f32(h_mul(a[i].m, b[i].s));
  }
  return sumf;
}

float co_vec_dot_q4_k_q8_k(const co_block_q4_k* a, const co_block_q8_k* b, size_t nb, int i16_wrap,
                           size_t* n_overflow) {
  const uint32_t KMASK1 = 0x3f3f3f3fu, KMASK2 = 0x0f0f0f0fu, KMASK3 = 0x03030303u;
  uint32_t utmp[4];
  int8_t aux8[256];
  int16_t aux16[8];
  float sums[8], aux32[8];
  memset(sums, 0, sizeof sums);
  float sumf = 0.0f;
  for (size_t bi = 0; bi < nb; bi++) {
    const uint8_t* q4 = a[bi].qs;
    const int8_t* q8 = b[bi].qs;
    memset(aux32, 0, sizeof aux32);
    for (int c = 0; c < 4; c++)
      for (int l = 0; l < 32; l++) {
        aux8[64 * c + l] = (int8_t)(q4[32 * c + l] & 0xF);
        aux8[64 * c + l + 32] = (int8_t)(q4[32 * c + l] >> 4);
      }
    for (int i = 0; i < 3; i++) {
      const uint8_t* s = a[bi].scales + 4 * i;
      utmp[i] = (uint32_t)s[0] | ((uint32_t)s[1] << 8) | ((uint32_t)s[2] << 16) | ((uint32_t)s[3] << 24);
    }
    utmp[3] = ((utmp[2] >> 4) & KMASK2) | (((utmp[1] >> 6) & KMASK3) << 4);
    uint32_t uaux = utmp[1] & KMASK1;
    utmp[1] = (utmp[2] & KMASK2) | (((utmp[0] >> 6) & KMASK3) << 4);
    utmp[2] = uaux;
    utmp[0] &= KMASK1;
    uint8_t scales[8], mins[8];
    for (int i = 0; i < 4; i++) {
      scales[i] = (uint8_t)(utmp[0] >> (8 * i));
      scales[4 + i] = (uint8_t)(utmp[1] >> (8 * i));
      mins[i] = (uint8_t)(utmp[2] >> (8 * i));
      mins[4 + i] = (uint8_t)(utmp[3] >> (8 * i));
    }
    int64_t sumi = 0;
    for (int j = 0; j < 16; j++) {
      int32_t prod = (int32_t)b[bi].bsums[j] * (int32_t)mins[j / 2];
      if (prod > 32767 || prod < -32768) {
        if (n_overflow) (*n_overflow)++;
        if (i16_wrap) prod = (int32_t)(int16_t)(uint16_t)((uint32_t)prod & 0xffffu);
      }
      sumi += prod;
    }
    for (int is = 0; is < 8; is++) {
      float scale = (float)scales[is];
      const int8_t* a8 = aux8 + 32 * is;
      const int8_t* b8 = q8 + 32 * is;
      for (int g = 0; g < 4; g++)
        for (int l = 0; l < 8; l++) {
          aux16[l] = (int16_t)((int16_t)b8[8 * g + l] * (int16_t)a8[8 * g + l]);
          aux32[l] += scale * (float)aux16[l];
        }
    }
    float d = co_f16_to_f32(a[bi].d) * b[bi].d;
    for (int l = 0; l < 8; l++) sums[l] += d * aux32[l];
    float dmin = co_f16_to_f32(a[bi].dmin) * b[bi].d;
    sumf -= dmin * (float)sumi;
  }
  for (int l = 0; l < 8; l++) sumf += sums[l];
  return sumf;
}

/* buf_q5_k.rs:229-325.  aux8 = the 5-bit levels (low nibble + 16 if the chunk's qh bit is set: mask m = 1 << (2c) for the first
 * 32 values of chunk c, 1 << (2c + 1) for the second); scales / mins unpacked as in Q4_K; eight f32 lanes (element e of a
 * 32-group feeds lane e % 8), `sums[l] += d * aux32[l]` and `sumf -= dmin * sumi` per super-block, the lanes added at the end. */
float co_vec_dot_q5_k_q8_k(const co_block_q5_k* a, const co_block_q8_k* b, size_t nb, int i16_wrap, size_t* n_overflow) {
  const uint32_t KMASK1 = 0x3f3f3f3fu, KMASK2 = 0x0f0f0f0fu, KMASK3 = 0x03030303u;
  uint32_t utmp[4];
  int8_t aux8[256];
  int16_t aux16[8];
  float sums[8], aux32[8];
  memset(sums, 0, sizeof sums);
  float sumf = 0.0f;
  for (size_t bi = 0; bi < nb; bi++) {
    const uint8_t* q5 = a[bi].qs;
    const uint8_t* qh = a[bi].qh;
    const int8_t* q8 = b[bi].qs;
    memset(aux32, 0, sizeof aux32);
    uint8_t m = 1;
    for (int c = 0; c < 4; c++) {
      for (int l = 0; l < 32; l++) aux8[64 * c + l] = (int8_t)((q5[32 * c + l] & 0xF) + ((qh[l] & m) ? 16 : 0));
      m = (uint8_t)(m << 1);
      for (int l = 0; l < 32; l++) aux8[64 * c + l + 32] = (int8_t)((q5[32 * c + l] >> 4) + ((qh[l] & m) ? 16 : 0));
      m = (uint8_t)(m << 1);
    }
    for (int i = 0; i < 3; i++) {
      const uint8_t* s = a[bi].scales + 4 * i;
      utmp[i] = (uint32_t)s[0] | ((uint32_t)s[1] << 8) | ((uint32_t)s[2] << 16) | ((uint32_t)s[3] << 24);
    }
    utmp[3] = ((utmp[2] >> 4) & KMASK2) | (((utmp[1] >> 6) & KMASK3) << 4);
    uint32_t uaux = utmp[1] & KMASK1;
    utmp[1] = (utmp[2] & KMASK2) | (((utmp[0] >> 6) & KMASK3) << 4);
    utmp[2] = uaux;
    utmp[0] &= KMASK1;
    uint8_t scales[8], mins[8];
    for (int i = 0; i < 4; i++) {
      scales[i] = (uint8_t)(utmp[0] >> (8 * i));
      scales[4 + i] = (uint8_t)(utmp[1] >> (8 * i));
      mins[i] = (uint8_t)(utmp[2] >> (8 * i));
      mins[4 + i] = (uint8_t)(utmp[3] >> (8 * i));
    }
    int64_t sumi = 0;
    for (int j = 0; j < 16; j++) {
      int32_t prod = (int32_t)b[bi].bsums[j] * (int32_t)mins[j / 2];
      if (prod > 32767 || prod < -32768) {
        if (n_overflow) (*n_overflow)++;
        if (i16_wrap) prod = (int32_t)(int16_t)(uint16_t)((uint32_t)prod & 0xffffu);
      }
      sumi += prod;
    }
    for (int is = 0; is < 8; is++) {
      float scale = (float)scales[is];
      const int8_t* a8 = aux8 + 32 * is;
      const int8_t* b8 = q8 + 32 * is;
      for (int g = 0; g < 4; g++)
        for (int l = 0; l < 8; l++) {
          aux16[l] = (int16_t)((int16_t)b8[8 * g + l] * (int16_t)a8[8 * g + l]);
          aux32[l] += scale * (float)aux16[l];
        }
    }
    float d = co_f16_to_f32(a[bi].d) * b[bi].d;
    for (int l = 0; l < 8; l++) sums[l] += d * aux32[l];
    float dmin = co_f16_to_f32(a[bi].dmin) * b[bi].d;
    sumf -= dmin * (float)sumi;
  }
  for (int l = 0; l < 8; l++) sumf += sums[l];
  return sumf;
}

float co_vec_dot_q6_k_q8_k(const co_block_q6_k* a, const co_block_q8_k* b, size_t nb) { /* buf_q6_k.rs:183-234 */
  int8_t aux8[256];
  int16_t aux16[8];
  float sums[8], aux32[8];
  for (int l = 0; l < 8; l++) sums[l] = 0.0f;
  for (size_t i = 0; i < nb; i++) {
    const uint8_t* q4b = a[i].ql;
    const uint8_t* qhb = a[i].qh;
    const int8_t* q8 = b[i].qs;
    for (int l = 0; l < 8; l++) aux32[l] = 0.0f;
    for (int j = 0; j < 256; j += 128) {
      int8_t* x8 = aux8 + j;
      const uint8_t* q4 = q4b + j / 2;
      const uint8_t* qh = qhb + j / 4;
      for (int l = 0; l < 32; l++) {
        x8[l] = (int8_t)((int32_t)((q4[l] & 0xF) | ((qh[l] & 3) << 4)) - 32);
        x8[l + 32] = (int8_t)((int32_t)((q4[l + 32] & 0xF) | (((qh[l] >> 2) & 3) << 4)) - 32);
        x8[l + 64] = (int8_t)((int32_t)((q4[l] >> 4) | (((qh[l] >> 4) & 3) << 4)) - 32);
        x8[l + 96] = (int8_t)((int32_t)((q4[l + 32] >> 4) | (((qh[l] >> 6) & 3) << 4)) - 32);
      }
    }
    for (int j = 0; j < 16; j++) {
      const float scale = (float)a[i].scales[j];
      const int8_t* q8j = q8 + 16 * j;
      const int8_t* a8j = aux8 + 16 * j;
      for (int l = 0; l < 8; l++) aux16[l] = (int16_t)((int16_t)q8j[l] * (int16_t)a8j[l]);
      for (int l = 0; l < 8; l++) aux32[l] += scale * (float)aux16[l];
      for (int l = 0; l < 8; l++) aux16[l] = (int16_t)((int16_t)q8j[8 + l] * (int16_t)a8j[8 + l]);
      for (int l = 0; l < 8; l++) aux32[l] += scale * (float)aux16[l];
    }
    const float d = co_f16_to_f32(a[i].d) * b[i].d;
    for (int l = 0; l < 8; l++) sums[l] += aux32[l] * d;
  }
  float sumf = 0.0f; /* Iterator::sum: left fold from 0.0 */
  for (int l = 0; l < 8; l++) sumf += sums[l];
  return sumf;
}

/* util.rs:29-152 (rmse_type 1 is all the Q6_K quantizer uses; the general function is restated) */
static float make_qx_quants(int n, int nmax, const float* data, int8_t* ls, int rmse_type) {
  float max = 0.0f, abs_max = 0.0f;
  for (int i = 0; i < n; i++) {
    float ax = fabsf(data[i]);
    if (ax > abs_max) {
      abs_max = ax;
      max = data[i];
    }
  }
  if (abs_max == 0.0f) {
    for (int i = 0; i < n; i++) ls[i] = 0;
    return 0.0f;
  }
  float iscale = -(float)nmax / max;
  if (rmse_type == 0) {
    for (int i = 0; i < n; i++) {
      int l = co_nearest_i32(iscale * data[i]);
      l = l < -nmax ? -nmax : (l > nmax - 1 ? nmax - 1 : l);
      ls[i] = (int8_t)(nmax + l);
    }
    return 1.0f / iscale;
  }
  const int weight_type = rmse_type % 2;
  float sumlx = 0.0f, suml2 = 0.0f;
  for (int i = 0; i < n; i++) {
    float xi = data[i];
    int l = co_nearest_i32(iscale * xi);
    l = l < -nmax ? -nmax : (l > nmax - 1 ? nmax - 1 : l);
    ls[i] = (int8_t)(l + nmax);
    float w = weight_type == 1 ? xi * xi : 1.0f;
    float lf = (float)l;
    sumlx += w * xi * lf;
    suml2 += w * lf * lf;
  }
  float scale = sumlx / suml2;
  float best = scale * sumlx;
  for (int itry = 0; itry < 3; itry++) {
    float isc = 1.0f / scale;
    float slx = 0.0f, sl2 = 0.0f;
    int changed = 0;
    for (int i = 0; i < n; i++) {
      float xi = data[i];
      int l = co_nearest_i32(isc * xi);
      l = l < -nmax ? -nmax : (l > nmax - 1 ? nmax - 1 : l);
      if (l + nmax != (int)ls[i]) changed = 1;
      float w = weight_type == 1 ? xi * xi : 1.0f;
      float lf = (float)l;
      slx += w * xi * lf;
      sl2 += w * lf * lf;
    }
    if (!changed || sl2 == 0.0f || slx * slx <= best * sl2) break;
    for (int i = 0; i < n; i++) {
      int l = co_nearest_i32(isc * data[i]);
      l = l < -nmax ? -nmax : (l > nmax - 1 ? nmax - 1 : l);
      ls[i] = (int8_t)(nmax + l);
    }
    sumlx = slx;
    suml2 = sl2;
    scale = sumlx / suml2;
    best = scale * sumlx;
  }
  for (int itry = 0; itry < 5; itry++) {
    int n_changed = 0;
    for (int i = 0; i < n; i++) {
      float xi = data[i];
      float w = weight_type == 1 ? xi * xi : 1.0f;
      int l = (int)ls[i] - nmax;
      float slx = sumlx - w * xi * (float)l;
      if (slx > 0.0f) {
        float sl2 = suml2 - w * (float)l * (float)l;
        int new_l = co_nearest_i32(xi * sl2 / slx);
        new_l = new_l < -nmax ? -nmax : (new_l > nmax - 1 ? nmax - 1 : new_l);
        if (new_l != l) {
          slx += w * xi * (float)new_l;
          sl2 += w * (float)new_l * (float)new_l;
          if (sl2 > 0.0f && slx * slx * suml2 > sumlx * sumlx * sl2) {
            ls[i] = (int8_t)(nmax + new_l);
            sumlx = slx;
            suml2 = sl2;
            scale = sumlx / suml2;
            best = scale * sumlx;
            n_changed++;
          }
        }
      }
    }
    if (n_changed == 0) break;
  }
  if (rmse_type < 3) return scale;
  for (int is = -4; is < 4; is++) {
    if (is == 0) continue;
    iscale = -((float)nmax + 0.1f * (float)is) / max;
    float slx = 0.0f, sl2 = 0.0f;
    for (int i = 0; i < n; i++) {
      float xi = data[i];
      int l = co_nearest_i32(iscale * xi);
      l = l < -nmax ? -nmax : (l > nmax - 1 ? nmax - 1 : l);
      float w = weight_type == 1 ? xi * xi : 1.0f;
      float lf = (float)l;
      slx += w * xi * lf;
      sl2 += w * lf * lf;
    }
    if (sl2 > 0.0f && slx * slx > best * sl2) {
      for (int i = 0; i < n; i++) {
        int l = co_nearest_i32(iscale * data[i]);
        l = l < -nmax ? -nmax : (l > nmax - 1 ? nmax - 1 : l);
        ls[i] = (int8_t)(nmax + l);
      }
      scale = slx / sl2;
      best = scale * slx;
    }
  }
  return scale;
}

void co_quantize_f32_q6_k(const float* x, size_t n, co_block_q6_k* out) { /* buf_q6_k.rs:109-181 */
  for (size_t bi = 0; bi < n / 256; bi++) {
    const float* chunk = x + bi * 256;
    int8_t l[256];
    float max_scale = 0.0f, max_abs_scale = 0.0f, scales[16];
    int8_t block_scales[16];
    uint8_t ql[128], qh[64];
    memset(l, 0, sizeof l);
    memset(ql, 0, sizeof ql);
    memset(qh, 0, sizeof qh);
    for (int ib = 0; ib < 16; ib++) {
      scales[ib] = make_qx_quants(16, 32, chunk + 16 * ib, l + 16 * ib, 1);
      float as = fabsf(scales[ib]);
      if (as > max_abs_scale) {
        max_abs_scale = as;
        max_scale = scales[ib];
      }
    }
    const float iscale = -128.0f / max_scale;
    const float d = 1.0f / iscale;
    for (int j = 0; j < 16; j++) {
      int v = co_nearest_i32(iscale * scales[j]);
      v = v < 127 ? v : 127;
      block_scales[j] = (int8_t)v; /* `as i8` of an i32 wraps */
    }
    for (int j = 0; j < 16; j++) {
      const float dj = d * (float)block_scales[j];
      if (dj == 0.0f) continue;
      for (int ii = 0; ii < 16; ii++) {
        int idx = 16 * j + ii;
        int ll = co_nearest_i32(chunk[idx] / dj);
        ll = ll < -32 ? -32 : (ll > 31 ? 31 : ll);
        l[idx] = (int8_t)(ll + 32);
      }
    }
    for (int j = 0; j < 256; j += 128) {
      int qi = j / 128;
      for (int li = 0; li < 32; li++) {
        int base = j + li;
        int8_t q1 = l[base] & 0xF, q2 = l[base + 32] & 0xF, q3 = l[base + 64] & 0xF, q4 = l[base + 96] & 0xF;
        ql[qi * 64 + li] = (uint8_t)(q1 | (q3 << 4));
        ql[qi * 64 + li + 32] = (uint8_t)(q2 | (q4 << 4));
        qh[qi * 32 + li] = (uint8_t)((l[base] >> 4) | ((l[base + 32] >> 4) << 2) | ((l[base + 64] >> 4) << 4) | ((l[base + 96] >> 4) << 6));
      }
    }
    memcpy(out[bi].ql, ql, 128);
    memcpy(out[bi].qh, qh, 64);
    memcpy(out[bi].scales, block_scales, 16);
    out[bi].d = co_f32_to_f16(d);
  }
}

float co_vec_dot_q8_k_q8_k(const co_block_q8_k* a, const co_block_q8_k* b, size_t nb) {
  float sumf = 0.0f;
  for (size_t i = 0; i < nb; i++) {
    int32_t s = 0;
    for (int j = 0; j < 256; j++) s += (int32_t)a[i].qs[j] * (int32_t)b[i].qs[j];
    sumf += (float)s * a[i].d * b[i].d;
  }
  return sumf;
}

float co_vec_dot_f32_f32(const float* a, const float* b, size_t n) {
  float sum = 0.0f;
  for (size_t i = 0; i < n; i++) sum += a[i] * b[i];
  return sum;
}

float co_vec_dot_f16_f16(const uint16_t* a, const uint16_t* b, size_t n) {
  float sum = 0.0f;
  for (size_t i = 0; i < n; i++) sum += co_f16_to_f32(a[i]) * co_f16_to_f32(b[i]);
  return sum;
}

/* ------------------------------------------------------------------------------------------
 * Dots, AVX2 lane order (archutil/x86_64.rs:6-53 helpers restated with the same intrinsics)
 * ---------------------------------------------------------------------------------------- */
int co_have_avx2(void) {
  return __builtin_cpu_supports("avx2") && __builtin_cpu_supports("fma") && __builtin_cpu_supports("f16c");
}

__attribute__((target("avx2,fma"))) static inline __m256 sum_i16_pairs_float(__m128i xh, __m128i xl) {
  __m128i ones = _mm_set1_epi16(1);
  __m128i l = _mm_madd_epi16(ones, xl);
  __m128i h = _mm_madd_epi16(ones, xh);
  return _mm256_cvtepi32_ps(_mm256_set_m128i(h, l));
}
__attribute__((target("avx2,fma"))) static inline __m256 mul_sum_us8_pairs_float(__m256i ax, __m256i sy) {
  __m128i axl = _mm256_castsi256_si128(ax), axh = _mm256_extractf128_si256(ax, 1);
  __m128i syl = _mm256_castsi256_si128(sy), syh = _mm256_extractf128_si256(sy, 1);
  __m128i dotl = _mm_maddubs_epi16(axl, syl), doth = _mm_maddubs_epi16(axh, syh);
  return sum_i16_pairs_float(doth, dotl);
}
__attribute__((target("avx2,fma"))) static inline __m256 mul_sum_i8_pairs_float(__m256i x, __m256i y) {
  __m256i ax = _mm256_sign_epi8(x, x);
  __m256i sy = _mm256_sign_epi8(y, x);
  return mul_sum_us8_pairs_float(ax, sy);
}
__attribute__((target("avx2,fma"))) static inline float hsum_float_8(__m256 x) {
  __m128 res = _mm256_extractf128_ps(x, 1);
  res = _mm_add_ps(res, _mm256_castps256_ps128(x));
  res = _mm_add_ps(res, _mm_movehl_ps(res, res));
  res = _mm_add_ss(res, _mm_movehdup_ps(res));
  return _mm_cvtss_f32(res);
}
__attribute__((target("avx2,fma"))) static inline __m256i bytes_from_nibbles_32(const uint8_t* p) {
  __m128i tmp = _mm_loadu_si128((const __m128i*)p);
  __m256i bytes = _mm256_set_m128i(_mm_srli_epi16(tmp, 4), tmp);
  return _mm256_and_si256(_mm256_set1_epi8(0xF), bytes);
}

__attribute__((target("avx2,fma"))) float co_vec_dot_q8_0_q8_0_avx2(const co_block_q8_0* a, const co_block_q8_0* b,
                                                                    size_t nb) {
  __m256 acc0 = _mm256_setzero_ps(), acc1 = _mm256_setzero_ps();
  size_t i = 0;
  for (; i + 1 < nb; i += 2) {
    __m256 d0 = _mm256_set1_ps(co_f16_to_f32(a[i].d) * co_f16_to_f32(b[i].d));
    __m256 d1 = _mm256_set1_ps(co_f16_to_f32(a[i + 1].d) * co_f16_to_f32(b[i + 1].d));
    __m256 q0 = mul_sum_i8_pairs_float(_mm256_loadu_si256((const __m256i*)a[i].qs),
                                       _mm256_loadu_si256((const __m256i*)b[i].qs));
    __m256 q1 = mul_sum_i8_pairs_float(_mm256_loadu_si256((const __m256i*)a[i + 1].qs),
                                       _mm256_loadu_si256((const __m256i*)b[i + 1].qs));
    acc0 = _mm256_fmadd_ps(d0, q0, acc0);
    acc1 = _mm256_fmadd_ps(d1, q1, acc1);
  }
  if (nb % 2 == 1) {
    const co_block_q8_0 *x = &a[nb - 1], *y = &b[nb - 1];
    __m256 d = _mm256_set1_ps(co_f16_to_f32(x->d) * co_f16_to_f32(y->d));
    __m256 q = mul_sum_i8_pairs_float(_mm256_loadu_si256((const __m256i*)x->qs),
                                      _mm256_loadu_si256((const __m256i*)y->qs));
    acc0 = _mm256_fmadd_ps(d, q, acc0);
  }
  return hsum_float_8(_mm256_add_ps(acc0, acc1));
}

__attribute__((target("avx2,fma"))) float co_vec_dot_q4_0_q8_0_avx2(const co_block_q4_0* a, const co_block_q8_0* b,
                                                                    size_t nb) {
  if (nb % 32 != 0) return co_vec_dot_q4_0_q8_0(a, b, nb); /* buf_q4_0.rs:220-223 */
  __m256 acc = _mm256_setzero_ps();
  for (size_t i = 0; i < nb; i++) {
    __m256 d = _mm256_set1_ps(co_f16_to_f32(a[i].d) * co_f16_to_f32(b[i].d));
    __m256i bx = bytes_from_nibbles_32(a[i].qs);
    bx = _mm256_sub_epi8(bx, _mm256_set1_epi8(8));
    __m256i by = _mm256_loadu_si256((const __m256i*)b[i].qs);
    __m256 q = mul_sum_i8_pairs_float(bx, by);
    acc = _mm256_fmadd_ps(d, q, acc);
  }
  return hsum_float_8(acc);
}

__attribute__((target("avx2,fma"))) float co_vec_dot_q8_k_q8_k_avx2(const co_block_q8_k* a, const co_block_q8_k* b,
                                                                    size_t nb) {
  __m256 acc = _mm256_setzero_ps();
  for (size_t i = 0; i < nb; i++) {
    __m256i sumi = _mm256_setzero_si256();
    for (int j = 0; j < 256; j += 32) {
      __m256i xs = _mm256_loadu_si256((const __m256i*)(a[i].qs + j));
      __m256i ys = _mm256_loadu_si256((const __m256i*)(b[i].qs + j));
      __m256i xs0 = _mm256_cvtepi8_epi16(_mm256_extracti128_si256(xs, 0));
      __m256i ys0 = _mm256_cvtepi8_epi16(_mm256_extracti128_si256(ys, 0));
      sumi = _mm256_add_epi32(sumi, _mm256_madd_epi16(xs0, ys0));
      __m256i xs1 = _mm256_cvtepi8_epi16(_mm256_extracti128_si256(xs, 1));
      __m256i ys1 = _mm256_cvtepi8_epi16(_mm256_extracti128_si256(ys, 1));
      sumi = _mm256_add_epi32(sumi, _mm256_madd_epi16(xs1, ys1));
    }
    __m256 d = _mm256_set1_ps(a[i].d * b[i].d);
    acc = _mm256_fmadd_ps(d, _mm256_cvtepi32_ps(sumi), acc);
  }
  return hsum_float_8(acc);
}

/* ------------------------------------------------------------------------------------------
 * Q5_0 / Q5_1 / Q2_K / Q3_K: the formats the reference handles with scalar code only
 * ---------------------------------------------------------------------------------------- */
float co_vec_dot_q5_0_q8_0(const co_block_q5_0* a, const co_block_q8_0* b, size_t nb) { /* buf_q5_0.rs:143-161 */
  float sumf = 0.0f;
  for (size_t i = 0; i < nb; i++) {
    const uint32_t qh = rd_u32le(a[i].qh);
    int32_t sumi = 0;
    for (int j = 0; j < 16; j++) {
      const uint32_t xh0 = ((qh & (1u << j)) >> j) << 4, xh1 = (qh & (1u << (j + 16))) >> (j + 12);
      const int32_t x0 = (((int32_t)a[i].qs[j] & 0x0F) | (int32_t)xh0) - 16, x1 = (((int32_t)a[i].qs[j] >> 4) | (int32_t)xh1) - 16;
      sumi += x0 * (int32_t)b[i].qs[j] + x1 * (int32_t)b[i].qs[j + 16];
    }
    sumf += (float)sumi * co_f16_to_f32(a[i].d) * co_f16_to_f32(b[i].d);
  }
  return sumf;
}
float co_vec_dot_q5_1_q8_1(const co_block_q5_1* a, const co_block_q8_1* b, size_t nb) { /* buf_q5_1.rs:141-160: f16 products, as Q4_1 */
  float sumf = 0.0f;
  for (size_t i = 0; i < nb; i++) {
    const uint32_t qh = rd_u32le(a[i].qh);
    int32_t sumi = 0;
    for (int j = 0; j < 16; j++) {
      const uint32_t xh0 = ((qh >> j) << 4) & 0x10, xh1 = (qh >> (j + 12)) & 0x10;
      const int32_t x0 = ((int32_t)a[i].qs[j] & 0xF) | (int32_t)xh0, x1 = ((int32_t)a[i].qs[j] >> 4) | (int32_t)xh1;
      sumi += x0 * (int32_t)b[i].qs[j] + x1 * (int32_t)b[i].qs[j + 16];
    }
    sumf += (float)sumi * co_f16_to_f32(h_mul(a[i].d, b[i].d)) + co_f16_to_f32(h_mul(a[i].m, b[i].s));
  }
  return sumf;
}
float co_vec_dot_q2_k_q8_k(const co_block_q2_k* a, const co_block_q8_k* b, size_t nb, int i16_wrap, size_t* n_overflow) { /* buf_q2_k.rs:216-258 */
  float sumf = 0.0f;
  size_t over = 0;
  for (size_t i = 0; i < nb; i++) {
    int32_t summs = 0;
    for (int j = 0; j < 16; j++) {
      int32_t prod = (int32_t)b[i].bsums[j] * (int32_t)(a[i].scales[j] >> 4);
      if (prod < -32768 || prod > 32767) over++;
      if (i16_wrap) prod = (int16_t)(uint16_t)((uint32_t)prod & 0xffffu);
      summs += prod;
      if (summs < -32768 || summs > 32767) over++;
      if (i16_wrap) summs = (int16_t)(uint16_t)((uint32_t)summs & 0xffffu);
    }
    const float dall = b[i].d * co_f16_to_f32(a[i].d), dmin = b[i].d * co_f16_to_f32(a[i].dmin);
    int32_t isum = 0;
    int is = 0;
    const int8_t* q8 = b[i].qs;
    for (int half = 0; half < 2; half++) {
      const uint8_t* q2 = a[i].qs + 32 * half;
      for (int shift = 0; shift < 8; shift += 2) {
        for (int h = 0; h < 2; h++) {
          const int32_t d = a[i].scales[is++] & 0xF;
          int32_t isuml = 0;
          for (int l = 16 * h; l < 16 * h + 16; l++) isuml += (int32_t)q8[l] * (int32_t)((q2[l] >> shift) & 3);
          isum += d * isuml;
        }
        q8 += 32;
      }
    }
    sumf += dall * (float)isum - dmin * (float)summs;
  }
  if (n_overflow) *n_overflow = over;
  return sumf;
}
/* aux8 of buf_q3_k.rs:247-281: the 256 signed 3-bit levels (2 low bits, minus 4 where the hmask bit is clear) */
static void q3k_levels(const co_block_q3_k* a, int8_t* aux8) {
  uint8_t m = 1;
  int o = 0;
  for (int half = 0; half < 2; half++) {
    const uint8_t* q3 = a->qs + 32 * half;
    for (int shift = 0; shift < 8; shift += 2) {
      for (int l = 0; l < 32; l++) aux8[o + l] = (int8_t)((int8_t)((q3[l] >> shift) & 3) - ((a->hmask[l] & m) ? 0 : 4));
      o += 32;
      m = (uint8_t)(m << 1);
    }
  }
}
float co_vec_dot_q3_k_q8_k(const co_block_q3_k* a, const co_block_q8_k* b, size_t nb) { /* buf_q3_k.rs:238-329 */
  float sums[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  int8_t aux8[256], sc[16];
  for (size_t i = 0; i < nb; i++) {
    q3k_levels(&a[i], aux8);
    q3k_scales(a[i].scales, sc);
    int32_t aux32[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    for (int j = 0; j < 16; j++) {
      const int32_t s = (int32_t)(int8_t)(sc[j] - 32);
      for (int g = 0; g < 2; g++)
        for (int l = 0; l < 8; l++) {
          const int16_t p = (int16_t)((int16_t)b[i].qs[16 * j + 8 * g + l] * (int16_t)aux8[16 * j + 8 * g + l]);
          aux32[l] += s * (int32_t)p;
        }
    }
    const float d = co_f16_to_f32(a[i].d) * b[i].d;
    for (int l = 0; l < 8; l++) sums[l] += d * (float)aux32[l];
  }
  float r = sums[0];
  for (int l = 1; l < 8; l++) r = r + sums[l];
  return r;
}

void co_quantize_f32_q5_0(const float* x, size_t n, co_block_q5_0* out) { /* buf_q5_0.rs:96-141 */
  for (size_t c = 0; c < n / 32; c++) {
    const float* ch = x + 32 * c;
    float max_val = 0.0f, max_abs = 0.0f;
    for (int i = 0; i < 32; i++) {
      const float av = fabsf(ch[i]);
      if (max_abs < av) {
        max_abs = av;
        max_val = ch[i];
      }
    }
    const float d = max_val / -16.0f;
    const float id = d != 0.0f ? 1.0f / d : 0.0f;
    uint32_t iqh = 0;
    for (int i = 0; i < 16; i++) {
      const float x0 = ch[i] * id, x1 = ch[i + 16] * id;
      int8_t a0 = rs_f32_as_i8(x0 + 16.5f), a1 = rs_f32_as_i8(x1 + 16.5f);
      const uint8_t xi0 = (uint8_t)(a0 < 31 ? a0 : 31), xi1 = (uint8_t)(a1 < 31 ? a1 : 31);
      out[c].qs[i] = (uint8_t)((xi0 & 0x0F) | ((xi1 & 0x0F) << 4));
      iqh |= (((uint32_t)xi0 & 0x10u) >> 4) << i;
      iqh |= (((uint32_t)xi1 & 0x10u) >> 4) << (i + 16);
    }
    for (int i = 0; i < 4; i++) out[c].qh[i] = (uint8_t)(iqh >> (8 * i));
    out[c].d = co_f32_to_f16(d);
  }
}
void co_quantize_f32_q5_1(const float* x, size_t n, co_block_q5_1* out) { /* buf_q5_1.rs:99-139 */
  for (size_t c = 0; c < n / 32; c++) {
    const float* ch = x + 32 * c;
    float mn = 3.40282347e+38f, mx = -3.40282347e+38f; /* f32::MAX, f32::MIN; f32::min / max skip NaN operands */
    for (int i = 0; i < 32; i++) {
      mn = fminf(ch[i], mn);
      mx = fmaxf(ch[i], mx);
    }
    const float d = (mx - mn) / 31.0f;
    const float id = d != 0.0f ? 1.0f / d : 0.0f;
    uint32_t iqh = 0;
    for (int i = 0; i < 16; i++) {
      const float x0 = (ch[i] - mn) * id, x1 = (ch[i + 16] - mn) * id;
      const uint8_t xi0 = rs_f32_as_u8(x0 + 0.5f), xi1 = rs_f32_as_u8(x1 + 0.5f);
      out[c].qs[i] = (uint8_t)((xi0 & 0x0F) | ((xi1 & 0x0F) << 4));
      iqh |= (((uint32_t)xi0 & 0x10u) >> 4) << i;
      iqh |= (((uint32_t)xi1 & 0x10u) >> 4) << (i + 16);
    }
    for (int i = 0; i < 4; i++) out[c].qh[i] = (uint8_t)(iqh >> (8 * i));
    out[c].d = co_f32_to_f16(d);
    out[c].m = co_f32_to_f16(mn);
  }
}
void co_quantize_f32_q2_k(const float* x, size_t n, co_block_q2_k* out) { /* buf_q2_k.rs:145-214 */
  uint8_t L[256]; /* declared outside the block loop in the reference: make_qkx1_quants' did_change sees the previous block's levels */
  float mins[16], scales[16];
  memset(L, 0, sizeof L);
  for (size_t i = 0; i < n / 256; i++) {
    const float* ch = x + 256 * i;
    co_block_q2_k* b = &out[i];
    memset(b, 0, sizeof *b);
    float max_scale = 0.0f, max_min = 0.0f;
    for (int j = 0; j < 16; j++) {
      scales[j] = make_qkx1_quants(16, 3, ch + 16 * j, L + 16 * j, &mins[j], 5);
      if (scales[j] > max_scale) max_scale = scales[j];
      if (mins[j] > max_min) max_min = mins[j];
    }
    if (max_scale > 0.0f) {
      const float iscale = 15.0f / max_scale;
      for (int j = 0; j < 16; j++) b->scales[j] = (uint8_t)(uint32_t)co_nearest_i32(iscale * scales[j]);
      b->d = co_f32_to_f16(max_scale / 15.0f);
    }
    if (max_min > 0.0f) {
      const float iscale = 15.0f / max_min;
      for (int j = 0; j < 16; j++) {
        const uint8_t l = (uint8_t)(uint32_t)co_nearest_i32(iscale * mins[j]);
        b->scales[j] |= (uint8_t)(l << 4);
      }
      b->dmin = co_f32_to_f16(max_min / 15.0f);
    }
    for (int j = 0; j < 16; j++) {
      const float d = co_f16_to_f32(b->d) * (float)(b->scales[j] & 0xF);
      if (d == 0.0f) continue;
      const float dm = co_f16_to_f32(b->dmin) * (float)(b->scales[j] >> 4);
      for (int ii = 0; ii < 16; ii++) {
        int32_t l = co_nearest_i32((x[16 * j + ii] + dm) / d); /* `data`, not `data_chunk`: buf_q2_k.rs:197 */
        l = l < 3 ? l : 3;
        l = l > 0 ? l : 0;
        L[16 * j + ii] = (uint8_t)l;
      }
    }
    for (int j = 0; j < 256; j += 128)
      for (int l = 0; l < 32; l++)
        b->qs[j / 4 + l] = (uint8_t)(L[j + l] | (L[j + l + 32] << 2) | (L[j + l + 64] << 4) | (L[j + l + 96] << 6));
  }
}
/* util.rs:218-284 (do_rmse = true is the only call: buf_q3_k.rs:167) */
static float make_q3_quants(int n, int nmax, const float* data, int8_t* l, int do_rmse) {
  float max = 0.0f, amax = 0.0f;
  for (int i = 0; i < n; i++) {
    const float ax = fabsf(data[i]);
    if (ax > amax) {
      amax = ax;
      max = data[i];
    }
  }
  if (amax == 0.0f) {
    for (int i = 0; i < n; i++) l[i] = 0;
    return 0.0f;
  }
  const float iscale = -(float)nmax / max;
  if (do_rmse) {
    float sumlx = 0.0f, suml2 = 0.0f;
    for (int i = 0; i < n; i++) {
      int32_t li = co_nearest_i32(iscale * data[i]);
      li = li < nmax - 1 ? li : nmax - 1;
      li = li > -nmax ? li : -nmax;
      l[i] = (int8_t)li;
      const float w = data[i] * data[i];
      sumlx += w * data[i] * (float)li;
      suml2 += w * (float)(li * li);
    }
    for (int t = 0; t < 5; t++) {
      int n_changed = 0;
      for (int i = 0; i < n; i++) {
        const float w = data[i] * data[i];
        float slx = sumlx - w * data[i] * (float)l[i];
        if (slx > 0.0f) {
          float sl2 = suml2 - w * (float)l[i] * (float)l[i];
          int32_t new_l = co_nearest_i32(data[i] * sl2 / slx);
          new_l = new_l < nmax - 1 ? new_l : nmax - 1;
          new_l = new_l > -nmax ? new_l : -nmax;
          if (new_l != (int32_t)l[i]) {
            slx += w * data[i] * (float)new_l;
            sl2 += w * (float)(new_l * new_l);
            if (sl2 > 0.0f && slx * slx * suml2 > sumlx * sumlx * sl2) {
              l[i] = (int8_t)new_l;
              sumlx = slx;
              suml2 = sl2;
              n_changed++;
            }
          }
        }
      }
      if (n_changed == 0) break;
    }
    for (int i = 0; i < n; i++) l[i] = (int8_t)(l[i] + nmax);
    return sumlx / suml2;
  }
  for (int i = 0; i < n; i++) {
    int32_t li = co_nearest_i32(iscale * data[i]);
    li = li < nmax - 1 ? li : nmax - 1;
    li = li > -nmax ? li : -nmax;
    l[i] = (int8_t)(li + nmax);
  }
  return 1.0f / iscale;
}
void co_quantize_f32_q3_k(const float* x, size_t n, co_block_q3_k* out) { /* buf_q3_k.rs:155-236 */
  int8_t L[256];
  float scales[16];
  for (size_t i = 0; i < n / 256; i++) {
    const float* ch = x + 256 * i;
    co_block_q3_k* b = &out[i];
    memset(b, 0, sizeof *b);
    float max_scale = 0.0f, amax = 0.0f;
    for (int j = 0; j < 16; j++) {
      scales[j] = make_q3_quants(16, 4, ch + 16 * j, L + 16 * j, 1);
      const float sc = fabsf(scales[j]);
      if (sc > amax) {
        amax = sc;
        max_scale = scales[j];
      }
    }
    if (max_scale != 0.0f) {
      const float iscale = -32.0f / max_scale;
      for (int j = 0; j < 16; j++) {
        int8_t l = rs_i32_as_i8(co_nearest_i32(iscale * scales[j]));
        l = (int8_t)((l < -32 ? -32 : l > 31 ? 31 : l) + 32);
        if (j < 8)
          b->scales[j] = (uint8_t)((uint8_t)l & 0xf);
        else
          b->scales[j - 8] |= (uint8_t)(((uint8_t)l & 0xf) << 4);
        l = (int8_t)(l >> 4);
        b->scales[j % 4 + 8] |= (uint8_t)((uint8_t)l << (2 * (j / 4)));
      }
      b->d = co_f32_to_f16(1.0f / iscale);
    }
    for (int j = 0; j < 16; j++) {
      int8_t sc = j < 8 ? (int8_t)(b->scales[j] & 0xf) : (int8_t)(b->scales[j - 8] >> 4);
      sc = (int8_t)((sc | (int8_t)(((b->scales[8 + j % 4] >> (2 * (j / 4))) & 3) << 4)) - 32);
      const float d = co_f16_to_f32(b->d) * (float)sc;
      if (d == 0.0f) continue;
      for (int ii = 0; ii < 16; ii++) {
        int32_t l = co_nearest_i32(ch[16 * j + ii] / d);
        l = l < -4 ? -4 : l > 3 ? 3 : l;
        L[16 * j + ii] = (int8_t)(l + 4);
      }
    }
    int m = 0;
    uint8_t hm = 1;
    for (int e = 0; e < 256; e++) {
      if (L[e] > 3) {
        b->hmask[m] |= hm;
        L[e] = (int8_t)(L[e] - 4);
      }
      if (++m == 32) {
        m = 0;
        hm = (uint8_t)(hm << 1);
      }
    }
    for (int j = 0; j < 256; j += 128)
      for (int l = 0; l < 32; l++)
        b->qs[j / 4 + l] = (uint8_t)(L[j + l] | (L[j + l + 32] << 2) | (L[j + l + 64] << 4) | (L[j + l + 96] << 6));
  }
}

/* ------------------------------------------------------------------------------------------
 * Exact integer parts (bit-exact gate for the HIP unpack + integer dot)
 * ---------------------------------------------------------------------------------------- */
int co_block_dots(const void* w, uint32_t wtype, const void* x, size_t n, int32_t* out) {
  size_t g = n / 32;
  switch (wtype) {
    case CO_Q4_0: {
      const co_block_q4_0* a = (const co_block_q4_0*)w;
      const co_block_q8_0* b = (const co_block_q8_0*)x;
      for (size_t i = 0; i < g; i++) {
        int32_t s = 0;
        for (int j = 0; j < 16; j++)
          s += ((int32_t)(a[i].qs[j] & 0xF) - 8) * b[i].qs[j] + ((int32_t)(a[i].qs[j] >> 4) - 8) * b[i].qs[j + 16];
        out[i] = s;
      }
      return 0;
    }
    case CO_Q8_0: {
      const co_block_q8_0* a = (const co_block_q8_0*)w;
      const co_block_q8_0* b = (const co_block_q8_0*)x;
      for (size_t i = 0; i < g; i++) {
        int32_t s = 0;
        for (int j = 0; j < 32; j++) s += (int32_t)a[i].qs[j] * b[i].qs[j];
        out[i] = s;
      }
      return 0;
    }
    case CO_Q4_1: {
      const co_block_q4_1* a = (const co_block_q4_1*)w;
      const co_block_q8_1* b = (const co_block_q8_1*)x;
      for (size_t i = 0; i < g; i++) {
        int32_t s = 0;
        for (int j = 0; j < 16; j++)
          s += (int32_t)(a[i].qs[j] & 0xF) * b[i].qs[j] + (int32_t)(a[i].qs[j] >> 4) * b[i].qs[j + 16];
        out[i] = s;
      }
      return 0;
    }
    case CO_Q4_K: {
      const co_block_q4_k* a = (const co_block_q4_k*)w;
      const co_block_q8_k* b = (const co_block_q8_k*)x;
      for (size_t i = 0; i < n / 256; i++)
        for (int c = 0; c < 4; c++) {
          int32_t lo = 0, hi = 0;
          for (int l = 0; l < 32; l++) {
            lo += (int32_t)(a[i].qs[32 * c + l] & 0xF) * b[i].qs[64 * c + l];
            hi += (int32_t)(a[i].qs[32 * c + l] >> 4) * b[i].qs[64 * c + 32 + l];
          }
          out[i * 8 + 2 * c] = lo;
          out[i * 8 + 2 * c + 1] = hi;
        }
      return 0;
    }
    case CO_Q5_K: {
      const co_block_q5_k* a = (const co_block_q5_k*)w;
      const co_block_q8_k* b = (const co_block_q8_k*)x;
      for (size_t i = 0; i < n / 256; i++)
        for (int c = 0; c < 4; c++) {
          int32_t lo = 0, hi = 0;
          for (int l = 0; l < 32; l++) {
            lo += ((int32_t)(a[i].qs[32 * c + l] & 0xF) + (((a[i].qh[l] >> (2 * c)) & 1) ? 16 : 0)) * b[i].qs[64 * c + l];
            hi += ((int32_t)(a[i].qs[32 * c + l] >> 4) + (((a[i].qh[l] >> (2 * c + 1)) & 1) ? 16 : 0)) * b[i].qs[64 * c + 32 + l];
          }
          out[i * 8 + 2 * c] = lo;
          out[i * 8 + 2 * c + 1] = hi;
        }
      return 0;
    }
    case CO_Q6_K: { /* per 16-element scale group: sum (q6 - 32) * q8 (16 per super-block) */
      const co_block_q6_k* a = (const co_block_q6_k*)w;
      const co_block_q8_k* b = (const co_block_q8_k*)x;
      for (size_t i = 0; i < n / 256; i++) {
        int8_t aux8[256];
        for (int j = 0; j < 256; j += 128) {
          const uint8_t* q4 = a[i].ql + j / 2;
          const uint8_t* qh = a[i].qh + j / 4;
          for (int l = 0; l < 32; l++) {
            aux8[j + l] = (int8_t)((int32_t)((q4[l] & 0xF) | ((qh[l] & 3) << 4)) - 32);
            aux8[j + l + 32] = (int8_t)((int32_t)((q4[l + 32] & 0xF) | (((qh[l] >> 2) & 3) << 4)) - 32);
            aux8[j + l + 64] = (int8_t)((int32_t)((q4[l] >> 4) | (((qh[l] >> 4) & 3) << 4)) - 32);
            aux8[j + l + 96] = (int8_t)((int32_t)((q4[l + 32] >> 4) | (((qh[l] >> 6) & 3) << 4)) - 32);
          }
        }
        for (int gq = 0; gq < 16; gq++) {
          int32_t sacc = 0;
          for (int l = 0; l < 16; l++) sacc += (int32_t)aux8[16 * gq + l] * b[i].qs[16 * gq + l];
          out[i * 16 + gq] = sacc;
        }
      }
      return 0;
    }
    case CO_Q5_0: {
      const co_block_q5_0* a = (const co_block_q5_0*)w;
      const co_block_q8_0* b = (const co_block_q8_0*)x;
      for (size_t i = 0; i < g; i++) {
        const uint32_t qh = rd_u32le(a[i].qh);
        int32_t s = 0;
        for (int j = 0; j < 16; j++) {
          const int32_t x0 = (int32_t)((a[i].qs[j] & 0xF) | (((qh >> j) & 1) << 4)) - 16;
          const int32_t x1 = (int32_t)((a[i].qs[j] >> 4) | (((qh >> (j + 16)) & 1) << 4)) - 16;
          s += x0 * b[i].qs[j] + x1 * b[i].qs[j + 16];
        }
        out[i] = s;
      }
      return 0;
    }
    case CO_Q5_1: {
      const co_block_q5_1* a = (const co_block_q5_1*)w;
      const co_block_q8_1* b = (const co_block_q8_1*)x;
      for (size_t i = 0; i < g; i++) {
        const uint32_t qh = rd_u32le(a[i].qh);
        int32_t s = 0;
        for (int j = 0; j < 16; j++) {
          const int32_t x0 = (int32_t)((a[i].qs[j] & 0xF) | (((qh >> j) & 1) << 4));
          const int32_t x1 = (int32_t)((a[i].qs[j] >> 4) | (((qh >> (j + 16)) & 1) << 4));
          s += x0 * b[i].qs[j] + x1 * b[i].qs[j + 16];
        }
        out[i] = s;
      }
      return 0;
    }
    case CO_Q2_K: { /* per 16-element scale group, in element order: sum q2 * q8 */
      const co_block_q2_k* a = (const co_block_q2_k*)w;
      const co_block_q8_k* b = (const co_block_q8_k*)x;
      for (size_t i = 0; i < n / 256; i++)
        for (int gq = 0; gq < 16; gq++) {
          const int half = gq / 8, shift = 2 * ((gq % 8) / 2), h = gq & 1;
          int32_t sacc = 0;
          for (int l = 0; l < 16; l++) sacc += (int32_t)((a[i].qs[32 * half + 16 * h + l] >> shift) & 3) * b[i].qs[16 * gq + l];
          out[i * 16 + gq] = sacc;
        }
      return 0;
    }
    case CO_Q3_K: { /* per 16-element scale group: sum (q3 - 4) * q8 */
      const co_block_q3_k* a = (const co_block_q3_k*)w;
      const co_block_q8_k* b = (const co_block_q8_k*)x;
      int8_t aux8[256];
      for (size_t i = 0; i < n / 256; i++) {
        q3k_levels(&a[i], aux8);
        for (int gq = 0; gq < 16; gq++) {
          int32_t sacc = 0;
          for (int l = 0; l < 16; l++) sacc += (int32_t)aux8[16 * gq + l] * b[i].qs[16 * gq + l];
          out[i * 16 + gq] = sacc;
        }
      }
      return 0;
    }
    case CO_Q8_K: {
      const co_block_q8_k* a = (const co_block_q8_k*)w;
      const co_block_q8_k* b = (const co_block_q8_k*)x;
      for (size_t i = 0; i < n / 256; i++)
        for (int c = 0; c < 8; c++) {
          int32_t s = 0;
          for (int l = 0; l < 32; l++) s += (int32_t)a[i].qs[32 * c + l] * b[i].qs[32 * c + l];
          out[i * 8 + c] = s;
        }
      return 0;
    }
    default: return -1;
  }
}

/* ------------------------------------------------------------------------------------------
 * exp / gelu tables
 * ---------------------------------------------------------------------------------------- */
void co_init_exp_cache(uint16_t* t) { /* cpu_device.rs:108-115 */
  for (uint32_t x = 0; x < 65536; x++) t[x] = co_f32_to_f16(expf(co_f16_to_f32((uint16_t)x)));
}
static float gelu_single(float x) { /* gelu.rs:19-22 */
  const float COEF_A = 0.044715f;
  const float S = (float)0.7978845608028654;
  return 0.5f * x * (1.0f + tanhf(S * x * (1.0f + COEF_A * x * x)));
}
void co_init_gelu_cache(uint16_t* t) { /* cpu_device.rs:117-124 */
  for (uint32_t x = 0; x < 65536; x++) t[x] = co_f32_to_f16(gelu_single(co_f16_to_f32((uint16_t)x)));
}
float co_exp_f32_cached(float x, const uint16_t* table) { /* buf_f32.rs:29-35 */
  return co_f16_to_f32(table[co_f32_to_f16(x)]);
}

/* ------------------------------------------------------------------------------------------
 * Device context + thread pool (thread_pool.rs:13-88: N workers + the caller runs thunk 0 and
 * busy-waits for the rest)
 * ---------------------------------------------------------------------------------------- */
typedef void (*co_job_fn)(void* arg, size_t job);

typedef struct {
  struct co_device* dev;
  int idx;
  pthread_t th;
} co_worker;

struct co_device {
  int thread_num;
  int use_avx2;
  uint16_t* exp_cache;
  uint16_t* gelu_cache;
  co_worker* workers;
  pthread_mutex_t mu;
  pthread_cond_t cv;
  uint64_t generation;
  int stop;
  co_job_fn fn;
  void* arg;
  size_t n_jobs;
  volatile long pending;
};

static void* worker_main(void* p) {
  co_worker* w = (co_worker*)p;
  struct co_device* d = w->dev;
  uint64_t seen = 0;
  for (;;) {
    pthread_mutex_lock(&d->mu);
    while (!d->stop && d->generation == seen) pthread_cond_wait(&d->cv, &d->mu);
    if (d->stop) {
      pthread_mutex_unlock(&d->mu);
      return NULL;
    }
    seen = d->generation;
    co_job_fn fn = d->fn;
    void* arg = d->arg;
    size_t n_jobs = d->n_jobs;
    pthread_mutex_unlock(&d->mu);
    /* jobs 1.. are dealt round-robin: job j -> worker (j-1) % thread_num  (thread_pool.rs:56-60) */
    for (size_t j = 1 + (size_t)w->idx; j < n_jobs; j += (size_t)d->thread_num) {
      fn(arg, j);
      __sync_fetch_and_sub(&d->pending, 1);
    }
  }
}

static void pool_run(struct co_device* d, co_job_fn fn, void* arg, size_t n_jobs) {
  if (n_jobs == 0) return;
  if (n_jobs > 1) {
    pthread_mutex_lock(&d->mu);
    d->fn = fn;
    d->arg = arg;
    d->n_jobs = n_jobs;
    d->pending = (long)n_jobs - 1;
    d->generation++;
    pthread_cond_broadcast(&d->cv);
    pthread_mutex_unlock(&d->mu);
  }
  fn(arg, 0);
  if (n_jobs > 1)
    while (d->pending > 0) _mm_pause(); /* busy loop, thread_pool.rs:66-70 */
}

co_device* co_device_new(int thread_num, int use_avx2) {
  if (thread_num < 1) thread_num = 1;
  struct co_device* d = (struct co_device*)calloc(1, sizeof *d);
  d->thread_num = thread_num;
  d->use_avx2 = use_avx2 && co_have_avx2();
  d->exp_cache = (uint16_t*)malloc(65536 * 2);
  co_init_exp_cache(d->exp_cache);
  d->gelu_cache = NULL;
  pthread_mutex_init(&d->mu, NULL);
  pthread_cond_init(&d->cv, NULL);
  d->workers = (co_worker*)calloc((size_t)thread_num, sizeof(co_worker));
  for (int i = 0; i < thread_num; i++) {
    d->workers[i].dev = d;
    d->workers[i].idx = i;
    pthread_create(&d->workers[i].th, NULL, worker_main, &d->workers[i]);
  }
  return d;
}

void co_device_free(co_device* d) {
  if (!d) return;
  pthread_mutex_lock(&d->mu);
  d->stop = 1;
  pthread_cond_broadcast(&d->cv);
  pthread_mutex_unlock(&d->mu);
  for (int i = 0; i < d->thread_num; i++) pthread_join(d->workers[i].th, NULL);
  free(d->workers);
  free(d->exp_cache);
  free(d->gelu_cache);
  pthread_mutex_destroy(&d->mu);
  pthread_cond_destroy(&d->cv);
  free(d);
}

const uint16_t* co_device_exp_cache(co_device* d) { return d->exp_cache; }

/* ------------------------------------------------------------------------------------------
 * matmul_vec (matmul_vec.rs:9-78)
 * ---------------------------------------------------------------------------------------- */
typedef struct {
  struct co_device* dev;
  const uint8_t* w;
  uint32_t wtype;
  size_t m, k, b;
  const void* xq; /* quantized activations, (b, k) */
  uint32_t xtype;
  float* c;
  size_t work_len, total;
} gemv_args;

static float vec_dot_dispatch(struct co_device* d, uint32_t wtype, const void* wrow, const void* xrow, size_t k) {
  switch (wtype) {
    case CO_F32: return co_vec_dot_f32_f32((const float*)wrow, (const float*)xrow, k);
    case CO_F16: return co_vec_dot_f16_f16((const uint16_t*)wrow, (const uint16_t*)xrow, k);
    case CO_Q8_0:
      return d->use_avx2 ? co_vec_dot_q8_0_q8_0_avx2((const co_block_q8_0*)wrow, (const co_block_q8_0*)xrow, k / 32)
                         : co_vec_dot_q8_0_q8_0((const co_block_q8_0*)wrow, (const co_block_q8_0*)xrow, k / 32);
    case CO_Q4_0:
      return d->use_avx2 ? co_vec_dot_q4_0_q8_0_avx2((const co_block_q4_0*)wrow, (const co_block_q8_0*)xrow, k / 32)
                         : co_vec_dot_q4_0_q8_0((const co_block_q4_0*)wrow, (const co_block_q8_0*)xrow, k / 32);
    case CO_Q4_1: /* the reference's AVX2 Q4_1 path is buggy (m + s, buf_q4_1.rs:235); fallback semantics only */
      return co_vec_dot_q4_1_q8_1((const co_block_q4_1*)wrow, (const co_block_q8_1*)xrow, k / 32);
    case CO_Q4_K:
      return co_vec_dot_q4_k_q8_k((const co_block_q4_k*)wrow, (const co_block_q8_k*)xrow, k / 256, 0, NULL);
    case CO_Q5_K: /* scalar only in the reference */
      return co_vec_dot_q5_k_q8_k((const co_block_q5_k*)wrow, (const co_block_q8_k*)xrow, k / 256, 0, NULL);
    case CO_Q5_0: return co_vec_dot_q5_0_q8_0((const co_block_q5_0*)wrow, (const co_block_q8_0*)xrow, k / 32);
    case CO_Q5_1: return co_vec_dot_q5_1_q8_1((const co_block_q5_1*)wrow, (const co_block_q8_1*)xrow, k / 32);
    case CO_Q2_K: return co_vec_dot_q2_k_q8_k((const co_block_q2_k*)wrow, (const co_block_q8_k*)xrow, k / 256, 0, NULL);
    case CO_Q3_K: return co_vec_dot_q3_k_q8_k((const co_block_q3_k*)wrow, (const co_block_q8_k*)xrow, k / 256);
    case CO_Q6_K: /* the reference has no SIMD path for Q6_K */
      return co_vec_dot_q6_k_q8_k((const co_block_q6_k*)wrow, (const co_block_q8_k*)xrow, k / 256);
    case CO_Q8_K:
      return d->use_avx2 ? co_vec_dot_q8_k_q8_k_avx2((const co_block_q8_k*)wrow, (const co_block_q8_k*)xrow, k / 256)
                         : co_vec_dot_q8_k_q8_k((const co_block_q8_k*)wrow, (const co_block_q8_k*)xrow, k / 256);
    default: return 0.0f / 0.0f;
  }
}

static void gemv_job(void* p, size_t job) {
  gemv_args* a = (gemv_args*)p;
  size_t begin = job * a->work_len;
  size_t end = begin + a->work_len;
  if (end > a->total) end = a->total;
  size_t wrow_bytes = a->k / co_block_elems(a->wtype) * co_block_bytes(a->wtype);
  size_t xrow_bytes = a->k / co_block_elems(a->xtype) * co_block_bytes(a->xtype);
  /* the reference walks 16-element chunks and derives (mi, bi) once per chunk
   * (matmul_vec.rs:64-71); per-element derivation is identical whenever a chunk does not
   * straddle a batch row, which always holds for b == 1 (the only case the runner issues). */
  for (size_t e = begin; e < end; e++) {
    size_t mi = e % a->m, bi = e / a->m;
    a->c[e] = vec_dot_dispatch(a->dev, a->wtype, a->w + mi * wrow_bytes, (const uint8_t*)a->xq + bi * xrow_bytes, a->k);
  }
}

int co_matmul_vec(co_device* d, const void* w, uint32_t wtype, size_t m, size_t k, const float* x, size_t b,
                  float* c) {
  uint32_t xtype = co_vec_dot_rhs_dtype(wtype);
  size_t be = co_block_elems(xtype);
  if (xtype == 0xffffffffu || be == 0 || k % be != 0 || k % co_block_elems(wtype) != 0) return -1;
  size_t xbytes = b * k / be * co_block_bytes(xtype);
  void* xq = malloc(xbytes ? xbytes : 1);
  /* single-threaded re-quantization of the whole rhs on every call (matmul_vec.rs:37-40) */
  if (co_quantize(x, b * k, xtype, xq) != 0) {
    free(xq);
    return -1;
  }
  gemv_args a;
  a.dev = d;
  a.w = (const uint8_t*)w;
  a.wtype = wtype;
  a.m = m;
  a.k = k;
  a.b = b;
  a.xq = xq;
  a.xtype = xtype;
  a.c = c;
  a.total = b * m;
  size_t tn = (size_t)d->thread_num;
  a.work_len = a.total / tn; /* matmul_vec.rs:45 */
  if (a.work_len == 0) a.work_len = a.total ? a.total : 1; /* (chunks_mut(0) would panic in the reference) */
  size_t n_jobs = (a.total + a.work_len - 1) / a.work_len;
  pool_run(d, gemv_job, &a, n_jobs);
  free(xq);
  return 0;
}

/* ------------------------------------------------------------------------------------------
 * batch_matmul (batch_matmul.rs:15-131)
 * ---------------------------------------------------------------------------------------- */
int co_batch_matmul(const float* a, size_t ba, size_t m, size_t k, const void* bdata, uint32_t btype, size_t bb,
                    size_t n, size_t sb0, size_t sb1, size_t sb2, float* c) {
  if (!(sb1 == 1 || sb2 == 1)) return -1;
  if (ba < bb) return -1;
  if (btype == CO_F32) { /* batch_matmul_naive_f32: accumulates onto the zeroed C; B batch = bi % bb */
    const float* b = (const float*)bdata;
    for (size_t i = 0; i < ba * m * n; i++) c[i] = 0.0f;
    for (size_t bi = 0; bi < ba; bi++)
      for (size_t mi = 0; mi < m; mi++)
        for (size_t ni = 0; ni < n; ni++)
          for (size_t ki = 0; ki < k; ki++)
            c[bi * (m * n) + mi * n + ni] += a[bi * (m * k) + mi * k + ki] * b[(bi % bb) * sb0 + ki * sb1 + ni * sb2];
    return 0;
  }
  if (btype != CO_F16) return -1;
  const uint16_t* b = (const uint16_t*)bdata;
  uint16_t* a16 = (uint16_t*)malloc((ba * m * k ? ba * m * k : 1) * 2);
  co_f32_to_f16_vec(a, a16, ba * m * k); /* batch_matmul.rs:39 */
  size_t bcast = ba / bb;
  if (sb1 == 1) { /* stride_bk == 1: vec_dot_f16_f16 (f32 accumulate) */
    for (size_t i = 0; i < ba * m * n; i++) {
      size_t ni = i % n;
      size_t mi = (i - ni) / n % m;
      size_t bia = (i - ni - mi * n) / (m * n);
      size_t oa = bia * (m * k) + mi * k;
      size_t ob = (bia / bcast) * sb0 + ni * sb2;
      c[i] = co_vec_dot_f16_f16(a16 + oa, b + ob, k);
    }
  } else { /* stride_bn == 1: vec_fma_f16_f16 into an f16 accumulator (buf_f16.rs:152-163) */
    uint16_t* tmpc = (uint16_t*)calloc(ba * m * n ? ba * m * n : 1, 2);
    for (size_t bia = 0; bia < ba; bia++)
      for (size_t mi = 0; mi < m; mi++)
        for (size_t ki = 0; ki < k; ki++) {
          uint16_t av = a16[bia * (m * k) + mi * k + ki];
          const uint16_t* brow = b + (bia / bcast) * sb0 + ki * sb1;
          uint16_t* crow = tmpc + bia * (m * n) + mi * n;
          for (size_t ni = 0; ni < n; ni++) crow[ni] = h_add(crow[ni], h_mul(brow[ni], av));
        }
    co_f16_to_f32_vec(tmpc, c, ba * m * n);
    free(tmpc);
  }
  free(a16);
  return 0;
}

/* ------------------------------------------------------------------------------------------
 * rms_norm (rms_norm.rs:9-47): per 32-chunk ordered reduce_sum, chunk sums added serially;
 * x /= sqrt(sum/len + eps) by true division.
 * ---------------------------------------------------------------------------------------- */
void co_rms_norm_inplace(float* x, size_t rows, size_t cols, float eps) {
  for (size_t r = 0; r < rows; r++) {
    float* v = x + r * cols;
    float sum = 0.0f;
    for (size_t c = 0; c + 32 <= cols; c += 32) {
      float s = -0.0f; /* simd_reduce_add_ordered(v, -0.0) */
      for (int j = 0; j < 32; j++) s += v[c + j] * v[c + j];
      sum += s;
    }
    float rms = sqrtf(sum / (float)cols + eps);
    for (size_t c = 0; c + 32 <= cols; c += 32)
      for (int j = 0; j < 32; j++) v[c + j] = v[c + j] / rms;
  }
}

/* rope.rs:10-80 */
static void rope_llama(float* buf, size_t len, size_t pos, size_t head_dim, size_t rope_dim) {
  float theta_scale = powf(10000.0f, -2.0f / (float)head_dim);
  for (size_t h = 0; h + head_dim <= len; h += head_dim) {
    float* chunk = buf + h;
    float theta = (float)pos;
    for (size_t i = 0; i < rope_dim; i += 2) {
      float c = cosf(theta), s = sinf(theta);
      theta *= theta_scale;
      float qp0 = chunk[i], qp1 = chunk[i + 1];
      chunk[i] = qp0 * c - qp1 * s;
      chunk[i + 1] = qp0 * s + qp1 * c;
    }
  }
}
static void rope_neox(float* buf, size_t len, size_t pos, size_t head_dim, size_t rope_dim) {
  for (size_t h = 0; h + head_dim <= len; h += head_dim) {
    float* chunk = buf + h;
    for (size_t i = 0; i < rope_dim / 2; i++) {
      float fe = 2.0f * (float)i / (float)head_dim;
      float timescale = powf(10000.0f, fe);
      float theta = (float)pos / timescale;
      float c = cosf(theta), s = sinf(theta);
      float qp0 = chunk[i], qp1 = chunk[i + head_dim / 2];
      chunk[i] = qp0 * c - qp1 * s;
      chunk[i + head_dim / 2] = qp0 * s + qp1 * c;
    }
  }
}
void co_rope_inplace(float* x, size_t n_batch, size_t bi_stride, size_t head_dim, int mode, size_t pos,
                     size_t rope_dim) {
  for (size_t bi = 0; bi < n_batch; bi++) {
    if (mode == 0)
      rope_llama(x + bi * bi_stride, bi_stride, pos + bi, head_dim, rope_dim);
    else
      rope_neox(x + bi * bi_stride, bi_stride, pos + bi, head_dim, rope_dim);
  }
}

/* softmax.rs:11-57 */
void co_softmax_inplace(co_device* d, float* x, size_t rows, size_t cols) {
  for (size_t r = 0; r < rows; r++) {
    float* v = x + r * cols;
    float max = -INFINITY;
    for (size_t i = 0; i < cols; i++) max = fmaxf(v[i], max);
    float sum = 0.0f;
    for (size_t i = 0; i < cols; i++) {
      v[i] = co_exp_f32_cached(v[i] - max, d->exp_cache);
      sum += v[i];
    }
    for (size_t i = 0; i < cols; i++) v[i] /= sum;
  }
}

void co_silu_inplace(co_device* d, float* x, size_t n) { /* silu.rs:6-13 */
  for (size_t i = 0; i < n; i++) {
    float nexp = co_exp_f32_cached(-x[i], d->exp_cache);
    x[i] /= 1.0f + nexp;
  }
}

void co_gelu_inplace(co_device* d, float* x, size_t n) { /* gelu.rs:11-17 */
  if (!d->gelu_cache) {
    d->gelu_cache = (uint16_t*)malloc(65536 * 2);
    co_init_gelu_cache(d->gelu_cache);
  }
  for (size_t i = 0; i < n; i++) x[i] = co_f16_to_f32(d->gelu_cache[co_f32_to_f16(x[i])]);
}

/* arithmetic.rs:5-68.  Faithful to chunks_exact(4): a tail of len%4 elements of `a` is left
 * untouched and `b` cycles over its floor(nb/4) whole chunks only. */
void co_add_inplace(float* a, size_t na, const float* b, size_t nb) {
  if (nb == 1) {
    for (size_t i = 0; i < na; i++) a[i] += b[0];
    return;
  }
  size_t bc = nb / 4;
  if (bc == 0) return;
  for (size_t ch = 0; ch < na / 4; ch++)
    for (int j = 0; j < 4; j++) a[4 * ch + j] = a[4 * ch + j] + b[4 * (ch % bc) + j];
}
void co_mul_inplace(float* a, size_t na, const float* b, size_t nb) {
  if (nb == 1) {
    for (size_t i = 0; i < na; i++) a[i] *= b[0];
    return;
  }
  size_t bc = nb / 4;
  if (bc == 0) return;
  for (size_t ch = 0; ch < na / 4; ch++)
    for (int j = 0; j < 4; j++) a[4 * ch + j] = a[4 * ch + j] * b[4 * (ch % bc) + j];
}

/* concatenate.rs:12-204 */
int co_concatenate(void* dst, uint32_t dtype, const size_t* ds, const size_t* dst_strides, const void* rhs,
                   uint32_t rtype, const size_t* rs, const size_t* rstrides, int ndim, int axis) {
  if (ndim < 1 || ndim > 3) return -1;
  if (!((dtype == CO_F32 && rtype == CO_F32) || (dtype == CO_F16 && rtype == CO_F16) ||
        (dtype == CO_F16 && rtype == CO_F32)))
    return -2; /* "can not concatenate {} and {}" */
  size_t sh[3] = {1, 1, 1}, st1[3] = {0, 0, 0}, st2[3] = {0, 0, 0};
  int off = 3 - ndim;
  for (int i = 0; i < ndim; i++) {
    sh[off + i] = rs[i];
    st1[off + i] = dst_strides[i];
    st2[off + i] = rstrides[i];
  }
  size_t base = (ndim == 1 ? ds[0] * dst_strides[0] : ds[axis] * dst_strides[axis]);
  for (size_t x = 0; x < sh[0]; x++)
    for (size_t y = 0; y < sh[1]; y++)
      for (size_t z = 0; z < sh[2]; z++) {
        size_t o1 = base + x * st1[0] + y * st1[1] + z * st1[2];
        size_t o2 = x * st2[0] + y * st2[1] + z * st2[2];
        if (dtype == CO_F32)
          ((float*)dst)[o1] = ((const float*)rhs)[o2];
        else if (rtype == CO_F16)
          ((uint16_t*)dst)[o1] = ((const uint16_t*)rhs)[o2];
        else
          ((uint16_t*)dst)[o1] = co_f32_to_f16(((const float*)rhs)[o2]);
      }
  return 0;
}

void co_contiguous(const void* src, void* dst, size_t es, const size_t* shape, const size_t* strides, int ndim) {
  size_t sh[3] = {1, 1, 1}, st[3] = {0, 0, 0};
  int off = 3 - ndim;
  for (int i = 0; i < ndim; i++) {
    sh[off + i] = shape[i];
    st[off + i] = strides[i];
  }
  size_t idx = 0;
  for (size_t i = 0; i < sh[0]; i++)
    for (size_t j = 0; j < sh[1]; j++)
      for (size_t k = 0; k < sh[2]; k++) {
        size_t o = i * st[0] + j * st[1] + k * st[2];
        memcpy((uint8_t*)dst + idx * es, (const uint8_t*)src + o * es, es);
        idx++;
      }
}

size_t co_argmax_last(const float* x, size_t n) { /* sampler.rs:109-116 */
  size_t best = 0;
  for (size_t i = 1; i < n; i++)
    if (!(x[i] < x[best])) best = i; /* max_by keeps the later element on Equal/unordered */
  return best;
}

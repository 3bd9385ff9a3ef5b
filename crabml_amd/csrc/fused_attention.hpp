// fused_attention.hpp -- attention kernels: one workgroup per head (decode), the long-context split, the row-tiled causal prefill kernel
// Part of the fused decode step (fused.hip includes the three fused_*.hpp files once, in order; they are not stand-alone
// translation units: the kernels are launched from fused.hip's host code).
#pragma once
#include "fused_common.hpp"

namespace crabml_hip {
// ---- attention: one workgroup per head -------------------------------------------------------------------
// batch_matmul.rs: f16 cache -> q rounded to f16, f32-accumulated QK^T in k order (buf_f16.rs:83-97),
// GQA head = h / (n_heads/n_kv); PV accumulated in f16 with a rounding after the product and after the sum
// (buf_f16.rs:152-163).  f32 cache -> plain f32 loops, kv head = h % n_kv (batch_matmul.rs:61-67).
// softmax.rs:36-54 with the f16 exp table; the row sum is sequential (bit-exact) up to 1024 positions and a
// block tree beyond that (documented tolerance 1e-6 relative).
// STAMP (tools/attn_lab.hip only): workgroup 0 records s_memtime at its phase boundaries into `stamps`
template <bool KV16, bool STAMP = false>
__global__ __launch_bounds__(256) void k_attn(const float* __restrict__ q, const void* __restrict__ kc,
                                              const void* __restrict__ vc, const int* __restrict__ pos_d,
                                              const unsigned short* __restrict__ exp_tab, float* __restrict__ out,
                                              signed char* __restrict__ xq, unsigned short* __restrict__ xd,
                                              void* __restrict__ xisum, int n_heads, int n_kv, int hd, int seq_cap,
                                              PrefetchPlan pf, int q81_in, long long* __restrict__ stamps = nullptr) {
  // q81_in: bits 0-7 = the output quantizer (0 Q8_0, 1 Q8_1), bit 8 = sequential softmax row sum at any length (strict device)
  const int q81 = q81_in & 255;
  const bool strict_sum = (q81_in & 256) != 0;
  if ((int)blockIdx.x >= n_heads) {
    prefetch_wg(pf, blockIdx.x - n_heads, gridDim.x - n_heads);
    return;
  }
  auto stamp = [&](int i) {
    if constexpr (STAMP) {
      if (blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x == 0) stamps[i] = (long long)__builtin_readcyclecounter();
    }
  };
  stamp(0);
  extern __shared__ float lds[];
  __shared__ float s_red[4];
  __shared__ float s_val;
  float* scores = lds;
  float* qs = lds + seq_cap;
  const int tid = threadIdx.x;
  const int head = blockIdx.x;
  const int kvh = KV16 ? head / (n_heads / n_kv) : head % n_kv;
  // blockIdx.y: row of a prefill batch = one more cached position per row (causal); 0 for a decode step
  q += (size_t)blockIdx.y * n_heads * hd;
  out += (size_t)blockIdx.y * n_heads * hd;
  // Position-independent loads go out first, so that their (cold, cross-XCD) latency overlaps the q staging
  // instead of adding two more serial round trips: the first 64 halves of the K row this thread will score
  // and the first 16 V values of the output column it will accumulate.  Rows past `seq` are read but unused.
  i32x4 kpre[8];
  const bool kp = KV16 && hd >= 64 && tid < seq_cap;
  if (kp) {
    const unsigned short* kr0 = (const unsigned short*)kc + ((size_t)kvh * seq_cap + tid) * hd;
#pragma unroll
    for (int u = 0; u < 8; u++) kpre[u] = *(const i32x4*)(kr0 + 8 * u);
  }
  unsigned short vpre[16];
  const bool vp = KV16 && tid < hd && seq_cap >= 16;
  if (vp) {
    const unsigned short* vr0 = (const unsigned short*)vc + (size_t)kvh * seq_cap * hd + tid;
#pragma unroll
    for (int u = 0; u < 16; u++) vpre[u] = vr0[(size_t)u * hd];
  }
  const int seq = *pos_d + 1 + (int)blockIdx.y;
  for (int i = tid; i < hd; i += blockDim.x) {
    float v = q[head * hd + i];
    qs[i] = KV16 ? h2f(f2h(v)) : v;  // quantize_f32_f16(bufa) (batch_matmul.rs:39)
  }
  __syncthreads();
  stamp(1);
  // ---- scores[t] = q . K[t]
  for (int t = tid; t < seq; t += blockDim.x) {
    float acc = 0.0f;
    if (KV16) {
      const unsigned short* kr = (const unsigned short*)kc + ((size_t)kvh * seq_cap + t) * hd;
      int i = 0;
      for (; i + 64 <= hd; i += 64) {  // 8 x 16-byte loads in flight; products still added in k order
        i32x4 kv[8];
        if (kp && t == tid && i == 0) {
#pragma unroll
          for (int u = 0; u < 8; u++) kv[u] = kpre[u];
        } else {
#pragma unroll
          for (int u = 0; u < 8; u++) kv[u] = *(const i32x4*)(kr + i + 8 * u);
        }
#pragma unroll
        for (int u = 0; u < 8; u++)
#pragma unroll
          for (int j = 0; j < 4; j++) {
            unsigned w = (unsigned)kv[u][j];
            acc += qs[i + 8 * u + 2 * j] * h2f((unsigned short)(w & 0xffffu));
            acc += qs[i + 8 * u + 2 * j + 1] * h2f((unsigned short)(w >> 16));
          }
      }
      for (; i + 8 <= hd; i += 8) {
        i32x4 kv = *(const i32x4*)(kr + i);
#pragma unroll
        for (int j = 0; j < 4; j++) {
          unsigned w = (unsigned)kv[j];
          acc += qs[i + 2 * j] * h2f((unsigned short)(w & 0xffffu));
          acc += qs[i + 2 * j + 1] * h2f((unsigned short)(w >> 16));
        }
      }
      for (; i < hd; i++) acc += qs[i] * h2f(kr[i]);
    } else {
      const float* kr = (const float*)kc + ((size_t)kvh * seq_cap + t) * hd;
      int i = 0;
      for (; i + 4 <= hd; i += 4) {
        f32x4 kv = *(const f32x4*)(kr + i);
        acc += qs[i] * kv[0];
        acc += qs[i + 1] * kv[1];
        acc += qs[i + 2] * kv[2];
        acc += qs[i + 3] * kv[3];
      }
      for (; i < hd; i++) acc += qs[i] * kr[i];
    }
    scores[t] = acc;
  }
  __syncthreads();
  stamp(2);
  // ---- softmax (in place; probabilities rounded to f16 for the f16 cache)
  softmax_row<KV16>(scores, seq, exp_tab, s_red, &s_val, strict_sum);
  stamp(3);
  // ---- out[n] = sum_t p[t] * V[t][n]
  float val = 0.0f;
  const int n = tid;
  if (n < hd) {
    if (KV16) {
      const unsigned short* vr = (const unsigned short*)vc + (size_t)kvh * seq_cap * hd + n;
      _Float16 c = (_Float16)0.0f;  // native f16 product and sum (devutil.hpp): the chain is one v_add_f16 per position
      int t = 0;
      for (; t + 16 <= seq; t += 16) {  // 16 loads in flight, then the (inherently serial) f16 accumulate chain
        unsigned short vv[16];
        if (vp && t == 0) {
#pragma unroll
          for (int u = 0; u < 16; u++) vv[u] = vpre[u];
        } else {
#pragma unroll
          for (int u = 0; u < 16; u++) vv[u] = vr[(size_t)(t + u) * hd];
        }
#pragma unroll
        for (int u = 0; u < 16; u++) {
          const _Float16 prod = hbits(vv[u]) * (_Float16)scores[t + u];  // scores hold f16-representable values
          c = c + prod;
        }
      }
      for (; t < seq; t++) {
        const _Float16 prod = hbits(vr[(size_t)t * hd]) * (_Float16)scores[t];
        c = c + prod;
      }
      val = (float)c;
    } else {
      const float* vr = (const float*)vc + (size_t)kvh * seq_cap * hd + n;
      float c = 0.0f;
      int t = 0;
      for (; t + 16 <= seq; t += 16) {
        float vv[16];
#pragma unroll
        for (int u = 0; u < 16; u++) vv[u] = vr[(size_t)(t + u) * hd];
#pragma unroll
        for (int u = 0; u < 16; u++) c += scores[t + u] * vv[u];
      }
      for (; t < seq; t++) c += scores[t] * vr[(size_t)t * hd];
      val = c;
    }
    out[head * hd + n] = val;
  }
  stamp(4);
  // ---- quantize the head's output for wo (only when blocks do not straddle heads)
  if (xq != nullptr) {
    const bool live = n < hd;  // hd % 32 == 0 here, so 32-lane groups are all-live or all-dead
    const float vq = live ? val : 0.f;
    const QLane o = q81 ? quant_lane32<true>(vq, live) : quant_lane32<false>(vq, live);
    if (live) {
      int e = head * hd + n;
      xq[e] = o.q;
      if ((n & 31) == 0) {
        xd[e >> 5] = o.d;
        if (q81)
          store_qaux<true>(xisum, e >> 5, o.aux);
        else
          store_qaux<false>(xisum, e >> 5, o.aux);
      }
    }
  }
  stamp(5);
}

// ---- attention at short context with K / V staged through LDS ---------------------------------------------------------
// k_attn above lets every thread fetch "its" K row / V column straight from the cache: 16-byte pieces at a 256-byte stride,
// in batches that each cost a memory round trip, all queued in order behind one another (s_memtime stamps at position 39 of
// the 8B shape, tools/attn_lab.hip: q staged 1.7 us | scores 2.2 | softmax 1.2 | PV 3.1 | quantize 0.4).  Here the
// workgroup copies the head's K and V rows [0, seq) into LDS with coalesced 16-byte loads that are ALL in flight at once --
// the first 64 rows are requested before the position is even known (rows past `seq` are allocated cache memory: read,
// never used) -- and the score / PV loops then run out of LDS.  Same arithmetic, element for element, as k_attn:
// q rounded to f16, f32 accumulation in k order (buf_f16.rs:83-97), softmax_row, f16-accumulated PV in position order
// (buf_f16.rs:152-163).  f16 cache, head_dim % 8 == 0; serves positions < S (the rows the launch has LDS for).
// K rows are padded by 16 bytes: a 16-lane group of ds_read_b128 then covers all 64 banks once (272 B = 68 dwords = 4 mod 64).
// q81 = 2: the output leaves as Q8_K planes (xq = quants, k8.d / k8.bs = scales / 16-element sums): the heads of a 256-element
// super-block exchange their columns as granules (q8k_exchange_store); head_dim in {64, 128, 256}.
struct AttnQ8K {
  Q8KExchange ex;
  float* d;
  short* bs;
  signed char* qp;  // the class-major copy of the quants
};
template <int HD, bool STAMP = false>  // HD: head_dim when known at compile time (128: the score loop is fully unrolled), 0 = run time
__global__ __launch_bounds__(256) void k_attn_s(const float* __restrict__ q, const unsigned short* __restrict__ kc,
                                                const unsigned short* __restrict__ vc, const int* __restrict__ pos_d,
                                                const unsigned short* __restrict__ exp_tab, float* __restrict__ out,
                                                signed char* __restrict__ xq, unsigned short* __restrict__ xd,
                                                void* __restrict__ xisum, int n_heads, int n_kv, int hd_rt, int seq_cap, int S,
                                                PrefetchPlan pf, int q81_in, long long* __restrict__ stamps, AttnQ8K k8) {
  const int q81 = q81_in & 255;  // bit 8: sequential softmax row sum at any length (strict device), as in k_attn
  const bool strict_sum = (q81_in & 256) != 0;
  if ((int)blockIdx.x >= n_heads) {
    prefetch_wg(pf, blockIdx.x - n_heads, gridDim.x - n_heads);
    return;
  }
  auto stamp = [&](int i) {
    if constexpr (STAMP) {
      if (blockIdx.x == 0 && threadIdx.x == 0) stamps[i] = (long long)__builtin_readcyclecounter();
    }
  };
  stamp(0);
  extern __shared__ float lds[];
  __shared__ float s_red[4];
  __shared__ float s_val;
  const int hd = HD ? HD : hd_rt;
  const int S4 = (S + 3) & ~3, kstr = hd + 8;
  float* scores = lds;
  float* qs = lds + S4;
  unsigned short* Ks = (unsigned short*)(qs + hd);
  unsigned short* Vs = Ks + (size_t)S * kstr;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int head = blockIdx.x;
  const int kvh = head / (n_heads / n_kv);
  const int ppr = hd >> 3;  // 16-byte pieces per row
  const i32x4* Kg = (const i32x4*)(kc + (size_t)kvh * seq_cap * hd);
  const i32x4* Vg = (const i32x4*)(vc + (size_t)kvh * seq_cap * hd);
  // ---- requests that do not depend on the position: q, and the first rows of K and V (4 pieces per thread each)
  constexpr int P0 = 4, P1 = 10;  // P0 * 256 pieces speculative (64 rows at head_dim 128), up to P1 * 256 more once seq is known
  const float qv = tid < hd ? q[head * hd + tid] : 0.f;
  const int spec = min(S * ppr, P0 * 256);
  i32x4 k0[P0], vb0[P0];
  // (unconditional, index clamped: a load inside a divergent branch gets its own s_waitcnt from the compiler -- four serial
  // round trips, ~4000 cycles, where all eight requests should be in flight together)
#pragma unroll
  for (int j = 0; j < P0; j++) {
    const int p = tid + 256 * j;
    const int pc = p < spec ? p : spec - 1;
    k0[j] = Kg[pc];
    vb0[j] = Vg[pc];
  }
  stamp(6);  // loads issued
  int seq = *pos_d + 1;
  seq = seq < S ? seq : S;  // (the host only launches this kernel for positions < S)
  if constexpr (STAMP) {
    if (seq < 0) stamps[15] = 0;  // (forces the wait for the scalar load here)
  }
  stamp(7);  // position known
  // quantize_f32_f16(bufa) (batch_matmul.rs:39): q is staged as f16 -- the score chain multiplies f16 by f16 (v_fma_mix_f32)
  unsigned short* q16 = (unsigned short*)qs;
  if (tid < hd) q16[tid] = f2h(qv);
  stamp(8);  // q arrived and staged
  // ---- the rest of the rows, now that seq is known (none at the short contexts this kernel is for): all in flight at once
  const int need = seq * ppr;
  for (int p0 = spec; p0 < need; p0 += P1 * 256) {
    i32x4 k1[P1], v1[P1];
#pragma unroll
    for (int j = 0; j < P1; j++) {
      const int p = p0 + tid + 256 * j;
      const int pc = p < need ? p : need - 1;  // clamped, not branched (see above)
      k1[j] = Kg[pc];
      v1[j] = Vg[pc];
    }
#pragma unroll
    for (int j = 0; j < P1; j++) {
      const int p = p0 + tid + 256 * j;
      if (p < need) {
        const int row = p / ppr, c = p - row * ppr;
        *(i32x4*)(Ks + (size_t)row * kstr + c * 8) = k1[j];
        *(i32x4*)(Vs + (size_t)row * hd + c * 8) = v1[j];
      }
    }
  }
#pragma unroll
  for (int j = 0; j < P0; j++) {
    const int p = tid + 256 * j;
    if (p < spec) {
      const int row = p / ppr, c = p - row * ppr;
      *(i32x4*)(Ks + (size_t)row * kstr + c * 8) = k0[j];
      *(i32x4*)(Vs + (size_t)row * hd + c * 8) = vb0[j];
    }
  }
  stamp(9);  // K / V pieces arrived and written to LDS (this wave)
  __syncthreads();
  stamp(1);
  // ---- scores[t] = q . K[t], f32 accumulation in k order (one cached position per thread)
  // One v_fma_mix_f32 per element: acc <- fl32(q[i] * k[i] + acc) with both factors taken as f16 straight from packed
  // registers.  That IS the reference's `acc += q[i] * k[i]` bit for bit: q (rounded to f16, batch_matmul.rs:39) and k carry
  // 11-bit significands, so the f32 product is exact and the fused operation rounds once where the separate multiply would
  // not have rounded at all.  (v_readlane + v_cvt + v_mul + v_add per element made the 128-element dot 3500 cycles of
  // instruction issue -- a lone wave per SIMD issues one instruction per ~5 cycles; the chain of 128 dependent operations is
  // ~1000.)  q comes out of LDS as broadcast 16-byte reads issued together with the K row's.
  typedef _Float16 h2q __attribute__((ext_vector_type(2)));
  auto score_of = [&](int t) -> float {
    const unsigned short* kr = Ks + (size_t)t * kstr;
    float acc = 0.0f;
    if constexpr (HD == 128) {
      i32x4 kv[16], qq[16];
#pragma unroll
      for (int i = 0; i < 16; i++) {
        kv[i] = *(const i32x4*)(kr + 8 * i);
        qq[i] = *(const i32x4*)(q16 + 8 * i);
      }
#pragma unroll
      for (int i = 0; i < 16; i++) {
#pragma unroll
        for (int j = 0; j < 4; j++) {
          const h2q kh = __builtin_bit_cast(h2q, (unsigned)kv[i][j]), qh = __builtin_bit_cast(h2q, (unsigned)qq[i][j]);
          acc = __builtin_fmaf((float)qh[0], (float)kh[0], acc);
          acc = __builtin_fmaf((float)qh[1], (float)kh[1], acc);
        }
      }
    } else {
      for (int i = 0; i < hd; i += 8) {
        const i32x4 kv = *(const i32x4*)(kr + i), qq = *(const i32x4*)(q16 + i);
#pragma unroll
        for (int j = 0; j < 4; j++) {
          const h2q kh = __builtin_bit_cast(h2q, (unsigned)kv[j]), qh = __builtin_bit_cast(h2q, (unsigned)qq[j]);
          acc = __builtin_fmaf((float)qh[0], (float)kh[0], acc);
          acc = __builtin_fmaf((float)qh[1], (float)kh[1], acc);
        }
      }
    }
    return acc;
  };
  if (seq <= 64) {
    // one wave holds the whole row: softmax.rs:36-54 without a barrier or an LDS round trip -- the same maximum, the same
    // table lookups, the same sequential sum (lanes past seq add +0.0), the same division and f16 rounding as softmax_row
    if (wave == 0) {
      const float sc = lane < seq ? score_of(lane) : -INFINITY;
      stamp(2);
      const float mx = wave_max_f32(sc);
      const float e = lane < seq ? exp_cached_f(sc - mx, exp_tab) : 0.0f;
      float sum = 0.0f;
#pragma unroll
      for (int i = 0; i < 64; i++) sum += rl_f(e, i);
      const float pv = e / sum;
      if (lane < seq) scores[lane] = h2f(f2h(pv));  // quantize_f32_f16 of the lhs (batch_matmul.rs:39)
    }
    __syncthreads();
  } else {
    for (int t = tid; t < seq; t += 256) scores[t] = score_of(t);
    __syncthreads();
    stamp(2);
    softmax_row<true>(scores, seq, exp_tab, s_red, &s_val, strict_sum);
  }
  stamp(3);
  // ---- out[n] = sum_t p[t] * V[t][n]: f16 product and f16 sum per position, in position order (buf_f16.rs:152-163).
  // A lane carries the chains of TWO adjacent output columns on packed f16 math (v_pk_mul_f16 + v_pk_add_f16 = the half
  // crate's product / sum roundings, one instruction pair per position for both columns): hd / 2 lanes, one LDS dword of V
  // and one broadcast probability per position.
  typedef _Float16 h2v __attribute__((ext_vector_type(2)));
  const int npair = hd >> 1;
  float v0 = 0.0f, v1 = 0.0f;
  if (tid < npair) {
    const unsigned* vr = (const unsigned*)Vs + tid;  // row stride hd / 2 dwords
    h2v c = {(_Float16)0.0f, (_Float16)0.0f};
    int t = 0;
    for (; t + 8 <= seq; t += 8) {
      unsigned vv[8];
      float pp[8];
#pragma unroll
      for (int u = 0; u < 8; u++) {
        vv[u] = vr[(size_t)(t + u) * npair];
        pp[u] = scores[t + u];
      }
#pragma unroll
      for (int u = 0; u < 8; u++) {
        const _Float16 ph = (_Float16)pp[u];  // scores hold f16-representable values
        const h2v p2 = {ph, ph};
        const h2v prod = __builtin_bit_cast(h2v, vv[u]) * p2;
        c = c + prod;
      }
    }
    for (; t < seq; t++) {
      const _Float16 ph = (_Float16)scores[t];
      const h2v p2 = {ph, ph};
      const h2v prod = __builtin_bit_cast(h2v, vr[(size_t)t * npair]) * p2;
      c = c + prod;
    }
    v0 = (float)c[0];
    v1 = (float)c[1];
    *(f32x2*)(out + head * hd + 2 * tid) = f32x2{v0, v1};
  }
  stamp(4);
  if (xq != nullptr && q81 == 2) {
    // Q8_K rhs for wo (Q4_K layers): the head's columns go to LDS (the q staging area is free by now), wave 0 exchanges
    // them with the other heads of the super-block and stores this head's share of the planes
    if (tid < npair) {
      qs[2 * tid] = v0;
      qs[2 * tid + 1] = v1;
    }
    __syncthreads();
    if (wave == 0) q8k_exchange_store(k8.ex, qs, head * hd, hd, lane, xq, k8.d, k8.bs, k8.qp);
  } else if (xq != nullptr && wave * 64 < npair) {
  // ---- quantize the head's output for wo: a 32-element block = the 16 lanes of one DPP row (two columns each)  // whole waves (hd % 32 == 0 here: rows of 16 lanes are all-live or all-dead)
    const bool live = tid < npair;
    const float a0 = live ? v0 : 0.f, a1 = live ? v1 : 0.f;
    const float amax = row16_max_f32(fmaxf(fabsf(a0), fabsf(a1)));
    const float dd = amax / 127.0f;
    int q0, q1, aux;
    if (!q81) {  // buf_q8_0.rs:87-134: q = trunc(x / d) (`as i8` of the i32 wraps), aux = the block's quant sum
      q0 = (int)(signed char)(unsigned char)((unsigned)rs_f32_as_i32(a0 / dd) & 0xffu);
      q1 = (int)(signed char)(unsigned char)((unsigned)rs_f32_as_i32(a1 / dd) & 0xffu);
      aux = row16_sum_i32(live ? q0 + q1 : 0);
    } else {  // buf_q8_1.rs:90-129: clamp, NaN -> -128, s = f16(d * sum q)
      q0 = (int)fminf(fmaxf(a0 / dd, -128.0f), 127.0f);
      q1 = (int)fminf(fmaxf(a1 / dd, -128.0f), 127.0f);
      const int sum = row16_sum_i32(live ? q0 + q1 : 0);
      aux = (int)f2h((float)sum * dd);
    }
    if (live) {
      const int e = head * hd + 2 * tid;
      *(unsigned short*)(xq + e) = (unsigned short)(((unsigned)q0 & 0xffu) | (((unsigned)q1 & 0xffu) << 8));
      if ((tid & 15) == 0) {
        xd[e >> 5] = f2h(dd);
        if (q81)
          store_qaux<true>(xisum, e >> 5, aux);
        else
          store_qaux<false>(xisum, e >> 5, aux);
      }
    }
  }
  stamp(5);
}
__host__ __device__ inline size_t attn_s_lds_bytes(int S, int hd) {
  return (size_t)(((S + 3) & ~3) + hd) * sizeof(float) + (size_t)S * (hd + 8) * 2 + (size_t)S * hd * 2;
}

// ---- attention at long context: the same arithmetic over every CU ----------------------------------------------
// One workgroup per head streams its whole K and V through one CU (~26 GB/s): 223 us per layer at 4000 cached
// positions.  From `attn_long_from` positions on the step uses three kernels instead (f16 cache, head_dim % 32
// == 0, n_heads / n_kv in {1, 2, 4, 8}); every rounding point and summation order is unchanged:
//   k_attn_scores   (kv head, 128-position split): each thread scores ONE cached position against the G q heads
//                   that share the kv head -- K is read once, f32 accumulation in k order (buf_f16.rs:83-97);
//   k_attn_softmax  (head): softmax_row over the score row, probabilities rounded to f16;
//   k_attn_pv       (kv head, 32-dim slice): V tiles are staged through LDS by the whole workgroup (read once for
//                   the G heads), and G x 16 lanes run the f16 chains, two dims per lane on packed f16 math
//                   (v_pk_mul_f16 + v_pk_add_f16 = the half crate's product / sum roundings, devutil.hpp).
template <int G>
__global__ __launch_bounds__(256) void k_attn_scores(const float* __restrict__ q, const unsigned short* __restrict__ kc,
                                                     const int* __restrict__ pos_d, float* __restrict__ scores_g,
                                                     int n_kv, int hd, int seq_cap, int nsplit, int row0) {
  // one thread per (cached position, q head of the group): the G threads of a position sit in adjacent lanes and
  // read the same K row (one fetch); each runs its own k-ordered f32 accumulation (buf_f16.rs:83-97)
  extern __shared__ float lds[];  // qs[G][hd]
  constexpr int TS = 256 / G;     // positions per workgroup
  const int tid = threadIdx.x;
  const int j = blockIdx.x / nsplit, sp = blockIdx.x % nsplit;
  // blockIdx.y: row of a prefill batch (row0 + y: one more cached position per row); a decode step has one row
  const int rowb = row0 + (int)blockIdx.y;
  const int seq = *pos_d + 1 + rowb;
  q += (size_t)rowb * n_kv * G * hd;
  scores_g += (size_t)blockIdx.y * n_kv * G * seq_cap;
  if (sp * TS >= seq) return;
  // q staged as f16 (quantize_f32_f16(bufa), batch_matmul.rs:39): the dot is one v_fma_mix_f32 per element -- the f32
  // product of two f16 values is exact, so fl32(q k + acc) is the reference's `acc += q * k` bit for bit (k_attn_s)
  unsigned short* q16 = (unsigned short*)lds;
  const int g = tid % G;
  const int t = sp * TS + tid / G;
  // the thread's K row does not depend on q: its first 128 bytes are requested before q is staged (rows past seq: clamped)
  const unsigned short* kr = kc + ((size_t)j * seq_cap + (t < seq ? t : seq - 1)) * hd;
  i32x4 kv0[8];
  if (hd >= 64) {
#pragma unroll
    for (int u = 0; u < 8; u++) kv0[u] = *(const i32x4*)(kr + 8 * u);
  }
  for (int idx = tid; idx < G * hd; idx += 256) {
    const int gq = idx / hd, i = idx - gq * hd;
    q16[idx] = f2h(q[(size_t)(j * G + gq) * hd + i]);
  }
  __syncthreads();
  if (t >= seq) return;
  const unsigned short* qg = q16 + g * hd;
  typedef _Float16 h2q __attribute__((ext_vector_type(2)));
  float acc = 0.0f;
  int i = 0;
  for (; i + 64 <= hd; i += 64) {  // 8 x 16-byte loads in flight; products still added in k order
    i32x4 kv[8], qq[8];
#pragma unroll
    for (int u = 0; u < 8; u++) {
      kv[u] = i == 0 ? kv0[u] : *(const i32x4*)(kr + i + 8 * u);
      qq[u] = *(const i32x4*)(qg + i + 8 * u);
    }
#pragma unroll
    for (int u = 0; u < 8; u++)
#pragma unroll
      for (int w4 = 0; w4 < 4; w4++) {
        const h2q kh = __builtin_bit_cast(h2q, (unsigned)kv[u][w4]), qh = __builtin_bit_cast(h2q, (unsigned)qq[u][w4]);
        acc = __builtin_fmaf((float)qh[0], (float)kh[0], acc);
        acc = __builtin_fmaf((float)qh[1], (float)kh[1], acc);
      }
  }
  for (; i + 8 <= hd; i += 8) {
    const i32x4 kv = *(const i32x4*)(kr + i), qq = *(const i32x4*)(qg + i);
#pragma unroll
    for (int w4 = 0; w4 < 4; w4++) {
      const h2q kh = __builtin_bit_cast(h2q, (unsigned)kv[w4]), qh = __builtin_bit_cast(h2q, (unsigned)qq[w4]);
      acc = __builtin_fmaf((float)qh[0], (float)kh[0], acc);
      acc = __builtin_fmaf((float)qh[1], (float)kh[1], acc);
    }
  }
  for (; i < hd; i++) acc = __builtin_fmaf(h2f(qg[i]), h2f(kr[i]), acc);
  scores_g[(size_t)(j * G + g) * seq_cap + t] = acc;
}

template <int NW>  // waves per workgroup: 4, or 16 for the decode step's one row per head
__global__ __launch_bounds__(NW * 64) void k_attn_softmax(const float* __restrict__ scores_g, const int* __restrict__ pos_d,
                                                         const unsigned short* __restrict__ exp_tab,
                                                         unsigned short* __restrict__ p16, int seq_cap, int row0, int strict_sum) {
  extern __shared__ float lds[];
  __shared__ float s_red[NW];
  __shared__ float s_val;
  const int head = blockIdx.x, seq = *pos_d + 1 + row0 + (int)blockIdx.y;
  scores_g += (size_t)blockIdx.y * gridDim.x * seq_cap;
  p16 += (size_t)blockIdx.y * gridDim.x * seq_cap;
  for (int t = threadIdx.x; t < seq; t += NW * 64) lds[t] = scores_g[(size_t)head * seq_cap + t];
  __syncthreads();
  softmax_row<true, NW>(lds, seq, exp_tab, s_red, &s_val, strict_sum != 0);
  for (int t = threadIdx.x; t < seq; t += NW * 64) p16[(size_t)head * seq_cap + t] = f2h(lds[t]);  // exact: already f16 values
}

typedef _Float16 h16x2 __attribute__((ext_vector_type(2)));
#define ATTN_PV_TILE 256
#define ATTN_PV_ROW (ATTN_PV_TILE + 4)  // words per LDS row: 16-byte aligned rows, shifted by 4 banks from each other
template <int G>
__global__ __launch_bounds__(256) void k_attn_pv(const unsigned short* __restrict__ p16, const unsigned short* __restrict__ vc,
                                                 const int* __restrict__ pos_d, float* __restrict__ out,
                                                 signed char* __restrict__ xq, unsigned short* __restrict__ xd,
                                                 void* __restrict__ xisum, int hd, int seq_cap, int q81, int row0) {
  constexpr int T = ATTN_PV_TILE, ROW = ATTN_PV_ROW;
  // LDS, two buffers each: V tile transposed to [16 dim pairs][T] words (a chain lane reads 4 consecutive positions
  // of its dim pair with one ds_read_b128), P tile [G][T] words holding {p, p} (the packed multiplier, ready-made)
  __shared__ __attribute__((aligned(16))) unsigned vt[2][16 * ROW];
  __shared__ __attribute__((aligned(16))) unsigned pt[2][G * ROW];
  const int tid = threadIdx.x;
  const int nslice = hd / 32;
  const int j = blockIdx.x / nslice, sl = blockIdx.x % nslice;
  const int seq = *pos_d + 1 + row0 + (int)blockIdx.y;
  {  // blockIdx.y: row of a prefill batch (xq is null there)
    const size_t n_heads = (size_t)(gridDim.x / nslice) * G;
    p16 += (size_t)blockIdx.y * n_heads * seq_cap;
    out += (size_t)(row0 + blockIdx.y) * n_heads * hd;
  }
  const unsigned short* vbase = vc + (size_t)j * seq_cap * hd + sl * 32;
  const int ntiles = (seq + T - 1) / T;
  // loader role (all threads): V piece = 16 B (4 dim pairs) of row (tid / 4) + 64 r, piece tid % 4;
  // P piece = 16 B = 8 positions of one head
  i32x4 vreg[4], preg;
  auto issue = [&](int tile) {
    const int t0 = tile * T;
#pragma unroll
    for (int r = 0; r < 4; r++) {
      int t = t0 + (tid >> 2) + 64 * r;
      t = t < seq_cap ? t : seq_cap - 1;  // rows past seq are read (inside the cache allocation) but never used
      vreg[r] = *(const i32x4*)(vbase + (size_t)t * hd + (tid & 3) * 8);
    }
    if (tid < G * (T / 8)) {
      const int g = tid / (T / 8), c8 = tid % (T / 8);
      int t = t0 + c8 * 8;
      t = t + 8 <= seq_cap ? t : seq_cap - 8;  // only past the end of the cache: those positions are never consumed
      preg = *(const i32x4*)(p16 + (size_t)(j * G + g) * seq_cap + t);
    }
  };
  auto commit = [&](int buf) {
#pragma unroll
    for (int r = 0; r < 4; r++) {
      const int tl = (tid >> 2) + 64 * r;
#pragma unroll
      for (int i = 0; i < 4; i++) vt[buf][((tid & 3) * 4 + i) * ROW + tl] = (unsigned)vreg[r][i];
    }
    if (tid < G * (T / 8)) {
      const int g = tid / (T / 8), c8 = tid % (T / 8);
      unsigned pp[8];
#pragma unroll
      for (int i = 0; i < 4; i++) {
        const unsigned w = (unsigned)preg[i];
        const unsigned a = w & 0xffffu, b = w >> 16;
        pp[2 * i] = a | (a << 16);
        pp[2 * i + 1] = b | (b << 16);
      }
      *(i32x4*)(&pt[buf][g * ROW + c8 * 8]) = i32x4{(int)pp[0], (int)pp[1], (int)pp[2], (int)pp[3]};
      *(i32x4*)(&pt[buf][g * ROW + c8 * 8 + 4]) = i32x4{(int)pp[4], (int)pp[5], (int)pp[6], (int)pp[7]};
    }
  };
  // chain role: lane c < G * 16 owns dims 2 dp, 2 dp + 1 of head j * G + g
  const bool chain = tid < G * 16;
  const int g = tid >> 4, dp = tid & 15;
  h16x2 c2 = {(_Float16)0.0f, (_Float16)0.0f};
  issue(0);
  commit(0);
  __syncthreads();
  for (int tile = 0; tile < ntiles; tile++) {
    const int buf = tile & 1;
    if (tile + 1 < ntiles) issue(tile + 1);
    if (chain) {
      const int nt = seq - tile * T < T ? seq - tile * T : T;
      const unsigned* vrow = &vt[buf][dp * ROW];
      const unsigned* prow = &pt[buf][g * ROW];
      int t = 0;
      // NB rounds of 8 positions: all the LDS reads of a round go out before its (serial) packed adds, so the LDS
      // latency is paid once per round
#define PV_ROUND(NB)                                                                                         \
  for (; t + 8 * NB <= nt; t += 8 * NB) {                                                                    \
    i32x4 vq[2 * NB], pq[2 * NB];                                                                            \
    _Pragma("unroll") for (int b = 0; b < 2 * NB; b++) {                                                     \
      vq[b] = *(const i32x4*)(vrow + t + 4 * b);                                                             \
      pq[b] = *(const i32x4*)(prow + t + 4 * b);                                                             \
    }                                                                                                        \
    _Pragma("unroll") for (int b = 0; b < 2 * NB; b++) _Pragma("unroll") for (int u = 0; u < 4; u++) {       \
      const h16x2 pr = __builtin_bit_cast(h16x2, (unsigned)vq[b][u]) * __builtin_bit_cast(h16x2, (unsigned)pq[b][u]); \
      c2 = c2 + pr;                                                                                          \
    }                                                                                                        \
  }
      PV_ROUND(4)
      PV_ROUND(1)
#undef PV_ROUND
      for (; t < nt; t++) {
        const h16x2 pr = __builtin_bit_cast(h16x2, vrow[t]) * __builtin_bit_cast(h16x2, prow[t]);
        c2 = c2 + pr;
      }
    }
    if (tile + 1 < ntiles) commit(buf ^ 1);  // the other buffer was last read one iteration ago (barrier below)
    __syncthreads();
  }
  if (!chain) return;
  const float v0 = (float)c2[0], v1 = (float)c2[1];
  const int head = j * G + g;
  const int e0 = head * hd + sl * 32 + 2 * dp;
  out[e0] = v0;
  out[e0 + 1] = v1;
  if (xq != nullptr) {  // the rhs block of the 32 dims held by this 16-lane DPP row (quant_lane32's arithmetic)
    const float amax = row16_max_f32(fmaxf(fabsf(v0), fabsf(v1)));
    const float dd = amax / 127.0f;
    int q0, q1;
    if (q81) {  // Q8_1 (buf_q8_1.rs:90-129)
      q0 = (int)fminf(fmaxf(v0 / dd, -128.0f), 127.0f);
      q1 = (int)fminf(fmaxf(v1 / dd, -128.0f), 127.0f);
    } else {  // Q8_0 (buf_q8_0.rs:87-134)
      q0 = (int)(signed char)(unsigned char)((unsigned)rs_f32_as_i32(v0 / dd) & 0xffu);
      q1 = (int)(signed char)(unsigned char)((unsigned)rs_f32_as_i32(v1 / dd) & 0xffu);
    }
    const int qs = row16_sum_i32(q0 + q1);
    xq[e0] = (signed char)q0;
    xq[e0 + 1] = (signed char)q1;
    if (dp == 0) {
      xd[e0 >> 5] = f2h(dd);
      if (q81)
        store_qaux<true>(xisum, e0 >> 5, (int)f2h((float)qs * dd));
      else
        store_qaux<false>(xisum, e0 >> 5, qs);
    }
  }
}


// ---- the PV pass with the products made by other waves ------------------------------------------------------------------
// A column's value is ONE f16 chain over the cached positions: c <- fl16(c + fl16(p_t * v_t)) (the half crate's product and
// sum roundings, in t order); that order is the result, so the pass cannot be split over positions, and its floor is the
// dependent-issue latency of the add: 10.4 cycles for v_pk_add_f16, 6-7 for v_add_f16 (tools/valu_chain_lab.hip).
// k_attn_pv's chain wave multiplies, adds packed pairs and waits for its own LDS reads: 21 cycles per position (37 us per
// layer at 4096 positions, Llama-3-8B).  Here a chain lane owns ONE column and only adds (v_add_f16, the high halves through
// SDWA selects): four producer waves read V and P straight from global memory (three tiles ahead, in registers), round the
// packed products and lay them out in LDS per column -- prod[column][t], rows of 2 T + 16 bytes so that the lanes'
// ds_read_b128 (eight positions) are conflict-free -- and the chain wave reads 64 positions ahead of its adds.  A workgroup
// carries two heads x 32 columns (64 chains = one wave; fewer chains per workgroup = less LDS traffic per position, which is
// what bounded the four-head version): grid = kv heads x 32-column slices x G / 2.  Measured 8-9 cycles per position + 5 us
// (profiles/r02_pv_lab_chain_breakdown.log).  Same products, same order as k_attn_pv: bit-identical (tests).
template <int G>
struct PvSplit {
  static constexpr int HG = G >= 2 ? 2 : 1;         // heads per workgroup
  static constexpr int NSUB = G / HG;
  static constexpr int T = 256;                     // positions per tile
  static constexpr int ROWB = T * 2 + 16;           // bytes per chain row
  static constexpr int CHAINS = HG * 32;
  static constexpr int THREADS = 5 * 64;            // one chain wave + four producer waves
  static constexpr int D = 3;                       // tiles of loads in flight per producer thread
  static constexpr size_t LDS = (size_t)2 * CHAINS * ROWB;
};
template <int G>
__global__ __launch_bounds__(320) void k_attn_pv_split(const unsigned short* __restrict__ p16, const unsigned short* __restrict__ vc,
                                                       const int* __restrict__ pos_d, float* __restrict__ out,
                                                       signed char* __restrict__ xq, unsigned short* __restrict__ xd,
                                                       void* __restrict__ xisum, int hd, int seq_cap, int q81, int row0) {
  typedef PvSplit<G> C;
  constexpr int T = C::T, ROWB = C::ROWB, CH = C::CHAINS, D = C::D, HG = C::HG, NSUB = C::NSUB;
  extern __shared__ __attribute__((aligned(16))) unsigned char prodb[];  // [2][CH][ROWB]
  const int tid = threadIdx.x;
  const int nslice = hd / 32;
  const int hsub = blockIdx.x % NSUB;
  const int j = blockIdx.x / NSUB / nslice, sl = blockIdx.x / NSUB % nslice;
  const int seq = *pos_d + 1 + row0 + (int)blockIdx.y;
  {  // blockIdx.y: row of a prefill batch (xq is null there)
    const size_t n_heads = (size_t)(gridDim.x / NSUB / nslice) * G;
    p16 += (size_t)blockIdx.y * n_heads * seq_cap;
    out += (size_t)(row0 + blockIdx.y) * n_heads * hd;
  }
  const int ntiles = (seq + T - 1) / T;
  // The producer loop is branch-free for whole groups of D tiles: a conditional issue or commit makes the compiler's
  // wait-count analysis assume the shortest path and drain ALL loads before every commit (measured: one memory round trip
  // per tile).  Loads for tiles past the last one are clamped reads inside the cache, never consumed.
  if (tid >= 64) {
    // ---- producer: work item = (8-column piece q of the slice, four consecutive positions) of each tile
    const int pt = tid - 64;
    const int q = pt & 3, tg = pt >> 2;
    const unsigned short* vbase = vc + (size_t)j * seq_cap * hd + sl * 32 + q * 8;
    const unsigned short* pbase = p16 + (size_t)(j * G + hsub * HG) * seq_cap;
    i32x4 vr[D][4];
    unsigned long long pr[D][HG];
    auto issue = [&](int tile, int s) {
      long long t0 = (long long)tile * T + 4 * tg;
      t0 = t0 + 4 <= seq_cap ? t0 : seq_cap - 4;  // past the end of the cache: never consumed
#pragma unroll
      for (int r = 0; r < 4; r++) vr[s][r] = *(const i32x4*)(vbase + (size_t)(t0 + r) * hd);
#pragma unroll
      for (int g = 0; g < HG; g++) pr[s][g] = *(const unsigned long long*)(pbase + (size_t)g * seq_cap + t0);
    };
    auto commit = [&](int buf, int s) {
      unsigned char* pb = prodb + (size_t)buf * CH * ROWB + 8 * tg;
#pragma unroll
      for (int g = 0; g < HG; g++) {
        unsigned pp[4];  // {p, p} of the four positions
        const unsigned lo = (unsigned)pr[s][g], hi = (unsigned)(pr[s][g] >> 32);
        pp[0] = (lo & 0xffffu) | (lo << 16);
        pp[1] = (lo >> 16) | (lo & 0xffff0000u);
        pp[2] = (hi & 0xffffu) | (hi << 16);
        pp[3] = (hi >> 16) | (hi & 0xffff0000u);
#pragma unroll
        for (int i = 0; i < 4; i++) {
          unsigned o[4];
#pragma unroll
          for (int r = 0; r < 4; r++)
            o[r] = __builtin_bit_cast(unsigned, __builtin_bit_cast(h16x2, (unsigned)vr[s][r][i]) * __builtin_bit_cast(h16x2, pp[r]));
          // column 2 i (low halves) and column 2 i + 1 (high halves) of the four positions
          const unsigned l0 = __builtin_amdgcn_perm(o[1], o[0], 0x05040100u), l1 = __builtin_amdgcn_perm(o[3], o[2], 0x05040100u);
          const unsigned h0 = __builtin_amdgcn_perm(o[1], o[0], 0x07060302u), h1 = __builtin_amdgcn_perm(o[3], o[2], 0x07060302u);
          const int rowi = g * 32 + q * 8 + 2 * i;
          *(unsigned long long*)(pb + (size_t)rowi * ROWB) = (unsigned long long)l0 | ((unsigned long long)l1 << 32);
          *(unsigned long long*)(pb + (size_t)(rowi + 1) * ROWB) = (unsigned long long)h0 | ((unsigned long long)h1 << 32);
        }
      }
    };
#pragma unroll
    for (int s = 0; s < D; s++) issue(s, s);
    commit(0, 0);
    issue(D, 0);
    __syncthreads();
    // tile k is consumed between barrier k and barrier k + 1 while tile k + 1 is committed into the other buffer.
    // Whole groups of D tiles run branch-free; the last ntiles % D tiles run guarded (a conditional commit makes the
    // compiler drain every outstanding load first -- harmless in the last one or two rounds, where nothing is left to fetch,
    // and it saves the one or two rounds of products nobody would read)
    int base = 0;
    for (; base + D <= ntiles; base += D) {
#pragma unroll
      for (int u = 0; u < D; u++) {
        const int tile = base + u;  // the chain wave consumes `tile`
        commit((tile + 1) & 1, (u + 1) % D);
        issue(tile + 1 + D, (u + 1) % D);
        __syncthreads();
      }
    }
#pragma unroll
    for (int u = 0; u < D - 1; u++) {
      const int tile = base + u;
      if (tile >= ntiles) break;
      if (tile + 1 < ntiles) commit((tile + 1) & 1, (u + 1) % D);
      __syncthreads();
    }
    return;
  }
  // ---- chain wave: lane c owns column sl * 32 + (c & 31) of head j * G + hsub * HG + (c >> 5)
  const bool chain = tid < CH;
  _Float16 c = (_Float16)0.0f;
  __syncthreads();
#define PV_ADD8(w)                                                  \
  _Pragma("unroll") for (int r_ = 0; r_ < 4; r_++) {                \
    const h16x2 p_ = __builtin_bit_cast(h16x2, (unsigned)(w)[r_]);  \
    c = c + p_[0];                                                  \
    c = c + p_[1];                                                  \
  }
  for (int tile = 0; tile < ntiles; tile++) {
    if (chain) {
      const int nt = seq - tile * T < T ? seq - tile * T : T;
      const unsigned char* row = prodb + (size_t)(tile & 1) * CH * ROWB + (size_t)tid * ROWB;
      int t = 0;
      if (nt >= 64) {
        i32x4 a[8], b[8];
#pragma unroll
        for (int u = 0; u < 8; u++) a[u] = *(const i32x4*)(row + 16 * u);
        for (; t + 128 <= nt; t += 128) {
#pragma unroll
          for (int u = 0; u < 8; u++) b[u] = *(const i32x4*)(row + 2 * (t + 64) + 16 * u);
#pragma unroll
          for (int u = 0; u < 8; u++) PV_ADD8(a[u])
          {  // unconditional (a branch here makes the waits for `b` drain these reads too); only used when t + 192 <= nt
            const int ta = t + 128 <= T - 64 ? t + 128 : T - 64;
#pragma unroll
            for (int u = 0; u < 8; u++) a[u] = *(const i32x4*)(row + 2 * ta + 16 * u);
          }
#pragma unroll
          for (int u = 0; u < 8; u++) PV_ADD8(b[u])
        }
        if (t + 64 <= nt) {  // `a` holds positions t .. t + 63
#pragma unroll
          for (int u = 0; u < 8; u++) PV_ADD8(a[u])
          t += 64;
        }
      }
      for (; t + 8 <= nt; t += 8) {
        const i32x4 v = *(const i32x4*)(row + 2 * t);
        PV_ADD8(v)
      }
      for (; t < nt; t++) c = c + *(const _Float16*)(row + 2 * t);
    }
    __syncthreads();
  }
#undef PV_ADD8
  if (!chain) return;
  const float v = (float)c;
  const int head = j * G + hsub * HG + (tid >> 5);
  const int e = head * hd + sl * 32 + (tid & 31);
  out[e] = v;
  if (xq != nullptr) {  // the rhs block of the 32 columns held by this half-wave (k_attn_pv's epilogue: quant_lane32's arithmetic)
    const QLane o = q81 ? quant_lane32<true>(v, true) : quant_lane32<false>(v, true);
    xq[e] = o.q;
    if ((tid & 31) == 0) {
      xd[e >> 5] = o.d;
      if (q81)
        store_qaux<true>(xisum, e >> 5, o.aux);
      else
        store_qaux<false>(xisum, e >> 5, o.aux);
    }
  }
}


// ---- fast-mode attention at long context: split-KV with f32 accumulation ("flash decoding") -------------------------------------
// The three kernels above reproduce the reference bit for bit, which pins the PV pass to ONE serial f16 chain per column
// (buf_f16.rs:152-163): 19 us per layer at 4096 cached positions for 16.8 MB of K / V (2.1 us at the HBM rate).  The FAST step
// (not the strict-order device, not CRABML_HIP_LLAMA_EXACT_ATTENTION) gives that exactness up from `attn_long_from` positions on,
// as SURVEY.md a18 planned: K and V are streamed ONCE by n_kv x S workgroups, each over its own slice of the cached positions,
//   scores  s_t = sum_i f16(q_i) * k_t,i               (q rounded to f16 as batch_matmul.rs:39 does; f32 accumulation, v_dot2)
//   local   m = max_t s_t,  e_t = exp(s_t - m),  l = sum_t e_t,  O = sum_t e_t * v_t        (all f32)
// and the LAST workgroup of a kv head to arrive (one returning atomicAdd per workgroup on the head's ticket word: nobody polls)
// merges the S partials (M = max m, out = sum O e^(m - M) / sum l e^(m - M)) and emits the head outputs + their Q8_0 / Q8_1
// blocks for wo.  What differs from the reference: (a) exp is the f32 function, not the f16 table of f16(s - max)
// (softmax.rs:43-53, buf_f32.rs:29-35), (b) the probabilities and the p * v products are not rounded to f16, (c) the sum over
// positions is f32 and tree-shaped.  Every one of these removes a rounding the reference makes: the result is CLOSER to exact
// arithmetic, and differs from the reference by the reference's own f16 noise (measured: DESIGN.md 2.2; asserted against the
// oracle at positions 224 ... 4095 in tests/test_hip_long_context_oracle.py).
// Lane mapping: a K / V row is HD f16 = LPR lanes x 16 bytes; a wave instruction covers RPI = 64 / LPR consecutive rows = one
// aligned 1 KiB request.  Each (wave, row class) keeps its own online-softmax state; they meet once, in LDS.
// Hand-off of the partials: agent-scope (write-through) stores -> vmcnt(0) -> barrier -> ticket; the last arriver reads them with
// agent-scope loads (MI355X_MICROARCH "handoff-flag", sc1 both sides).  The ticket word only ever grows (S per launch).
template <int HD>
__device__ __forceinline__ float flash_row_sum(float v) {
  constexpr int LPR = HD / 8;
  v += dpp_f<0xB1>(v);
  v += dpp_f<0x4E>(v);
  v += dpp_f<0x141>(v);
  if constexpr (LPR == 16) v += dpp_f<0x140>(v);
  return v;
}
#define FLASH_MIN_ROWS 64  /* measured on MI355X, 8B shape: 64 <= 128 <= 256 at every context (profiles/r04_flash_sweep.log) */
#define FLASH_MAX_SLICES 32
template <int G>
struct FlashGeom {
  static constexpr int NW = G == 8 ? 4 : 8;  // waves per workgroup (LDS: NW * RPI classes x G x HD floats)
};
__host__ __device__ inline size_t flash_lds_bytes(int G, int hd) {
  const int nw = G == 8 ? 4 : 8, rpi = 64 / (hd / 8), ncls = nw * rpi;
  return (size_t)ncls * G * hd * 4 + (size_t)ncls * G * 4 * 3 + 64;
}
__host__ __device__ inline size_t flash_part_floats(int G, int hd) { return (size_t)G * (hd + 2); }
template <int G, int HD, bool Q81, bool TICKET>
__global__ __launch_bounds__(FlashGeom<G>::NW * 64) void k_attn_flash(const float* __restrict__ q, const unsigned short* __restrict__ kc,
                                                                     const unsigned short* __restrict__ vc, const int* __restrict__ pos_d,
                                                                     float* __restrict__ part, unsigned* __restrict__ tick,
                                                                     float* __restrict__ out, signed char* __restrict__ xq,
                                                                     unsigned short* __restrict__ xd, void* __restrict__ xisum, int seq_cap,
                                                                     int Smax, int min_rows) {
  constexpr int NW = FlashGeom<G>::NW, LPR = HD / 8, RPI = 64 / LPR, NCLS = NW * RPI, U = 4, NT = NW * 64;
  static_assert(HD == 64 || HD == 128, "a K / V row is 8 or 16 lanes x 16 bytes");
  typedef _Float16 h2 __attribute__((ext_vector_type(2)));
  extern __shared__ __attribute__((aligned(16))) float fl_lds[];
  float* sacc = fl_lds;                 // [NCLS][G][HD]
  float* sm = sacc + NCLS * G * HD;     // [NCLS][G]
  float* sl = sm + NCLS * G;            // [NCLS][G]
  float* sw = sl + NCLS * G;            // [NCLS][G] merge weights
  int* s_last = (int*)(sw + NCLS * G);
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int rc = lane / LPR, dl = lane % LPR;
  const int j = blockIdx.x / Smax, sp = blockIdx.x % Smax;
  // q of the G heads that share kv head j: this lane's 8 dims (requested before the position is read: the two round trips overlap)
  f32x4 qa[G], qb[G];
#pragma unroll
  for (int g = 0; g < G; g++) {
    const f32x4* qp = (const f32x4*)(q + (size_t)(j * G + g) * HD + 8 * dl);
    qa[g] = qp[0];
    qb[g] = qp[1];
  }
  const int seq = *pos_d + 1;
  // The grid is sized for the longest context (Smax slices per kv head); a step uses S <= Smax of them -- at least min_rows
  // (FLASH_MIN_ROWS) cached rows per workgroup: every slice costs the merge G x HD partial values, and a short context does not
  // repay many of them.  The other workgroups leave at once (every workgroup derives the same S from seq).
  const int S = min(Smax, max(1, (seq + min_rows - 1) / min_rows));
  if (sp >= S) return;
  // this workgroup's rows [r0, r1): the cache cut into S slices of whole row groups
  const int chunk = (((seq + S - 1) / S + RPI - 1) / RPI) * RPI;
  const int r0 = sp * chunk, r1 = r0 + chunk < seq ? r0 + chunk : seq;
  const int ngroups = r1 > r0 ? (r1 - r0 + RPI - 1) / RPI : 0;
  // rounded to f16 (batch_matmul.rs:39)
  i32x4 qh[G];
#pragma unroll
  for (int g = 0; g < G; g++) {
    const f32x4 a = qa[g], b = qb[g];
    qh[g] = i32x4{(int)((unsigned)f2h(a[0]) | ((unsigned)f2h(a[1]) << 16)), (int)((unsigned)f2h(a[2]) | ((unsigned)f2h(a[3]) << 16)),
                  (int)((unsigned)f2h(b[0]) | ((unsigned)f2h(b[1]) << 16)), (int)((unsigned)f2h(b[2]) | ((unsigned)f2h(b[3]) << 16))};
  }
  float m[G], l[G], acc[G][8];
#pragma unroll
  for (int g = 0; g < G; g++) {
    m[g] = -INFINITY;
    l[g] = 0.f;
#pragma unroll
    for (int i = 0; i < 8; i++) acc[g][i] = 0.f;
  }
  const unsigned short* kb = kc + (size_t)j * seq_cap * HD + 8 * dl;
  const unsigned short* vb = vc + (size_t)j * seq_cap * HD + 8 * dl;
  for (int g0 = wave; g0 < ngroups; g0 += NW * U) {
    i32x4 kk[U], vv[U];
#pragma unroll
    for (int u = 0; u < U; u++) {  // every load of the round is in flight before the first dot
      const int row = r0 + (g0 + u * NW) * RPI + rc;
      const int rowc = row < r1 ? row : r1 - 1;
      kk[u] = __builtin_nontemporal_load((const i32x4*)(kb + (size_t)rowc * HD));
      vv[u] = __builtin_nontemporal_load((const i32x4*)(vb + (size_t)rowc * HD));
    }
#pragma unroll
    for (int u = 0; u < U; u++) {
      if (g0 + u * NW >= ngroups) break;  // wave-uniform
      const bool live = r0 + (g0 + u * NW) * RPI + rc < r1;
      float vf[8];
#pragma unroll
      for (int i = 0; i < 4; i++) {
        const h2 t = __builtin_bit_cast(h2, (unsigned)vv[u][i]);
        vf[2 * i] = (float)t[0];
        vf[2 * i + 1] = (float)t[1];
      }
#pragma unroll
      for (int g = 0; g < G; g++) {
        float s = 0.f;
#pragma unroll
        for (int i = 0; i < 4; i++)
          s = __builtin_amdgcn_fdot2(__builtin_bit_cast(h2, (unsigned)kk[u][i]), __builtin_bit_cast(h2, (unsigned)qh[g][i]), s, false);
        s = flash_row_sum<HD>(s);
        s = live ? s : -INFINITY;
        const float mn = fmaxf(m[g], s);
        // (a class that has seen nothing yet and sees a dead row keeps m = -inf: alpha = 1, p = 0)
        const float alpha = mn == -INFINITY ? 1.0f : __expf(m[g] - mn);
        const float pe = mn == -INFINITY ? 0.0f : __expf(s - mn);
        m[g] = mn;
        l[g] = l[g] * alpha + pe;
#pragma unroll
        for (int i = 0; i < 8; i++) acc[g][i] = __builtin_fmaf(pe, vf[i], acc[g][i] * alpha);
      }
    }
  }
  // ---- the NCLS (wave, row class) states meet in LDS ---------------------------------------------------------------
  const int cls = wave * RPI + rc;
#pragma unroll
  for (int g = 0; g < G; g++) {
    f32x4* dst = (f32x4*)(sacc + ((size_t)cls * G + g) * HD + 8 * dl);
    dst[0] = f32x4{acc[g][0], acc[g][1], acc[g][2], acc[g][3]};
    dst[1] = f32x4{acc[g][4], acc[g][5], acc[g][6], acc[g][7]};
    if (dl == 0) {
      sm[cls * G + g] = m[g];
      sl[cls * G + g] = l[g];
    }
  }
  __syncthreads();
  if (tid < NCLS * G) {  // merge weights e^(m_c - M) per (class, head); tid = c * G + g
    const int g = tid % G;
    float M = -INFINITY;
    for (int c = 0; c < NCLS; c++) M = fmaxf(M, sm[c * G + g]);
    const float mc = sm[tid];
    sw[tid] = mc == -INFINITY ? 0.0f : __expf(mc - M);
  }
  __syncthreads();
  float* mypart = part + (size_t)blockIdx.x * (G * (HD + 2));
  for (int idx = tid; idx < G * HD; idx += NT) {
    const int g = idx / HD, d = idx % HD;
    float o = 0.f;
#pragma unroll 8
    for (int c = 0; c < NCLS; c++) o = __builtin_fmaf(sacc[((size_t)c * G + g) * HD + d], sw[c * G + g], o);
    if constexpr (TICKET)
      __hip_atomic_store(mypart + g * (HD + 2) + d, o, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    else
      mypart[g * (HD + 2) + d] = o;
  }
  if (tid < G) {
    float M = -INFINITY, L = 0.f;
    for (int c = 0; c < NCLS; c++) M = fmaxf(M, sm[c * G + tid]);
    for (int c = 0; c < NCLS; c++) L = __builtin_fmaf(sl[c * G + tid], sw[c * G + tid], L);
    if constexpr (TICKET) {
      __hip_atomic_store(mypart + tid * (HD + 2) + HD, M, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __hip_atomic_store(mypart + tid * (HD + 2) + HD + 1, L, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    } else {
      mypart[tid * (HD + 2) + HD] = M;
      mypart[tid * (HD + 2) + HD + 1] = L;
    }
  }
  if constexpr (!TICKET) return;  // the merge is its own launch (k_attn_flash_merge)
  // ---- ticket: the last workgroup of kv head j to get here merges the S partials ----------------------------------
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // this thread's write-through stores have been acknowledged
  __syncthreads();
  if (tid == 0) {
    const unsigned old = __hip_atomic_fetch_add(tick + j, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const int last = old == (unsigned)(S - 1) ? 1 : 0;
    // the last arriver re-arms the word for the next launch (every other workgroup of this kv head has drawn its ticket)
    if (last) __hip_atomic_store(tick + j, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    *s_last = last;
  }
  __syncthreads();
  if (*s_last == 0) return;
  // ---- merge (one workgroup per kv head): every load of the hand-off is requested before anything waits on one --------------
  const float* pj = part + (size_t)j * Smax * (G * (HD + 2));
  float* mg_m = sacc;                                  // [S][G]  (sacc is done with)
  float* mg_l = sacc + FLASH_MAX_SLICES * G;           // [S][G]
  float* mg_w = sacc + 2 * FLASH_MAX_SLICES * G;       // [S][G] e^(m_s - M)
  float* mg_M = sacc + 3 * FLASH_MAX_SLICES * G;       // [G] {1 / L}
  constexpr int SB = 16;                               // partial values per thread and batch
  auto load_o = [&](int idx, int s0, float (&ov)[SB]) {
    const int g = idx / HD, d = idx % HD;
#pragma unroll
    for (int i = 0; i < SB; i++) {
      const int s2 = s0 + i < S ? s0 + i : S - 1;
      ov[i] = __hip_atomic_load(pj + (size_t)s2 * (G * (HD + 2)) + g * (HD + 2) + d, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
  };
  float ov[SB];
  if (tid < G * HD) load_o(tid, 0, ov);
  for (int t = tid; t < S * G; t += NT) {  // the slices' {m, l} -> LDS
    const float* ps = pj + (size_t)(t / G) * (G * (HD + 2)) + (t % G) * (HD + 2) + HD;
    mg_m[t] = __hip_atomic_load(ps, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    mg_l[t] = __hip_atomic_load(ps + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  __syncthreads();
  if (tid < G) {  // per head: M, the slices' weights e^(m_s - M), L
    float M = -INFINITY, L = 0.f;
    for (int s2 = 0; s2 < S; s2++) M = fmaxf(M, mg_m[s2 * G + tid]);
    for (int s2 = 0; s2 < S; s2++) {
      const float ms = mg_m[s2 * G + tid];
      const float w = ms == -INFINITY ? 0.0f : __expf(ms - M);
      mg_w[s2 * G + tid] = w;
      L = __builtin_fmaf(mg_l[s2 * G + tid], w, L);
    }
    mg_M[tid] = L;
  }
  __syncthreads();
  for (int idx = tid; idx < G * HD; idx += NT) {  // (whole waves stay together: 64 | G * HD; one pass unless G * HD > NT)
    const int g = idx / HD, d = idx % HD;
    if (idx != tid) load_o(idx, 0, ov);
    float o = 0.f;
#pragma unroll
    for (int i = 0; i < SB; i++)
      if (i < S) o = __builtin_fmaf(ov[i], mg_w[i * G + g], o);
    for (int s0 = SB; s0 < S; s0 += SB) {  // contexts past SB slices: one more batch
      load_o(idx, s0, ov);
#pragma unroll
      for (int i = 0; i < SB; i++)
        if (s0 + i < S) o = __builtin_fmaf(ov[i], mg_w[(s0 + i) * G + g], o);
    }
    const float L = mg_M[g];
    const float val = o / L;
    const int e = (j * G + g) * HD + d;
    out[e] = val;
    if (xq != nullptr) {  // the rhs block of wo: 32 consecutive dims = one half-wave (quant_lane32's arithmetic)
      const QLane ql = quant_lane32<Q81>(val, true);
      xq[e] = ql.q;
      if ((lane & 31) == 0) {
        xd[e >> 5] = ql.d;
        store_qaux<Q81>(xisum, e >> 5, ql.aux);
      }
    }
  }
}

// The merge as its own launch (the default): one workgroup per q head, one thread per output dim.  Reads the S partials the
// k_attn_flash launch before it left behind (plain loads: the kernel boundary orders them), in slice order.
template <int HD, bool Q81>
__global__ __launch_bounds__(HD) void k_attn_flash_merge(const float* __restrict__ part, const int* __restrict__ pos_d,
                                                        float* __restrict__ out, signed char* __restrict__ xq,
                                                        unsigned short* __restrict__ xd, void* __restrict__ xisum, int G, int Smax,
                                                        int min_rows) {
  __shared__ float s_w[FLASH_MAX_SLICES];
  __shared__ float s_L;
  const int head = blockIdx.x, j = head / G, g = head % G, d = threadIdx.x;
  const float* pj = part + ((size_t)j * Smax * G + g) * (HD + 2);  // slice s of this head: pj + s * G * (HD + 2)
  const size_t sstr = (size_t)G * (HD + 2);
  // every slot of the grid is requested at once, whatever the position says (slots past S hold an earlier step's finite
  // values and get weight 0): the loads do not wait for the position's round trip
  float ov[FLASH_MAX_SLICES];
#pragma unroll
  for (int i = 0; i < FLASH_MAX_SLICES; i++) ov[i] = pj[(size_t)(i < Smax ? i : Smax - 1) * sstr + d];
  const int sl = d < Smax ? d : Smax - 1;  // lanes 0 .. Smax - 1 of wave 0 own one slice's {m, l}
  float ms = pj[(size_t)sl * sstr + HD], ls = pj[(size_t)sl * sstr + HD + 1];
  const int seq = *pos_d + 1;
  const int S = min(Smax, max(1, (seq + min_rows - 1) / min_rows));  // k_attn_flash's own rule
  if (d < 64) {  // wave 0: M, the slices' weights e^(m_s - M), L -- the k_attn_flash merge's operations in its order
    if (d >= S) ms = -INFINITY;
    const float M = wave_max_f32(ms);
    const float w = ms == -INFINITY ? 0.0f : __expf(ms - M);
    if (d < FLASH_MAX_SLICES) s_w[d] = w;
    float L = 0.f;
    for (int s2 = 0; s2 < S; s2++) L = __builtin_fmaf(rl_f(ls, s2), rl_f(w, s2), L);
    if (d == 0) s_L = L;
  }
  __syncthreads();
  float o = 0.f;
#pragma unroll
  for (int i = 0; i < FLASH_MAX_SLICES; i++)
    if (i < S) o = __builtin_fmaf(ov[i], s_w[i], o);
  const float val = o / s_L;
  const int e = head * HD + d;
  out[e] = val;
  if (xq != nullptr) {
    const QLane ql = quant_lane32<Q81>(val, true);
    xq[e] = ql.q;
    if ((d & 31) == 0) {
      xd[e >> 5] = ql.d;
      store_qaux<Q81>(xisum, e >> 5, ql.aux);
    }
  }
}

// ---- fast-mode attention of the batched prefill: causal flash attention on the f16 matrix cores ---------------------------------
// The prompt pass's exact attention (k_attn_tile / the three long-context kernels with a row dimension) runs the reference's f16
// chains per (row, head, column): 137 us per layer at 512 prompt rows, most of a 4096-token prompt.  The FAST step (same switch and the
// same stated deviation as k_attn_flash: f32 exp and accumulation; additionally the probabilities enter the second
// matrix product as f16, as the reference's do) runs it as one launch on v_mfma_f32_16x16x32_f16 / 16x16x16_f16:
//   workgroup = (64 prompt rows, head); wave = 16 of the rows; per 16 cached positions (two such chunks per step)
//     S^T[pos][row] = K[pos][:] . Q[row][:]      A = 16 K rows (lane: position l % 16, dims 32 ks + 8 (l / 16) .. + 8: one 16-byte
//                                                LDS read), B = the wave's q rows, rounded to f16 once (batch_matmul.rs:39)
//     online softmax per q row = per C column n = l % 16: a lane holds positions 4 (l / 16) .. + 4 of its row; the running maximum
//                                                is agreed across the four lane groups (two cross-lane steps), the row sum stays
//                                                per lane until the end; causal mask p <= pos0 + row on the C registers
//     O^T[dim][row] += V^T[dim][pos] . P^T[pos][row]   B = the S^T registers themselves (exp'd, packed to f16): the C layout of one
//                                                product IS the B layout of the next, no transpose; A = V^T from an LDS tile that the
//                                                workgroup fills TRANSPOSED
//   K and V tiles of 64 positions go through LDS (two buffers each, one barrier per fill): the next tile's loads are requested
//   before the current one is multiplied, so their round trip runs under a whole tile of matrix work.
// GQA: the G heads of a kv head are separate workgroups (their K / V reads meet in L2).
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
#define FLASH_ROWS_BN 64                   // cached positions per V^T fill
#define FLASH_ROWS_VSTR (FLASH_ROWS_BN + 4)  // halfs per V^T row: 8-byte aligned fragments, rows 34 banks apart
#define FLASH_ROWS_KPAD 8                   // halfs of padding per K row in LDS (16-byte aligned fragments, rows 4 banks apart)
__host__ __device__ inline size_t flash_rows_lds_bytes(int hd) {
  return (size_t)2 * ((size_t)hd * FLASH_ROWS_VSTR + (size_t)FLASH_ROWS_BN * (hd + FLASH_ROWS_KPAD)) * 2;
}
template <int HD>
__global__ __launch_bounds__(512) void k_attn_flash_rows(const float* __restrict__ q, const unsigned short* __restrict__ kc,
                                                         const unsigned short* __restrict__ vc, const int* __restrict__ pos_d,
                                                         float* __restrict__ out, int n_heads, int n_kv, int seq_cap, int B) {
  constexpr int KS = HD / 32, DT = HD / 16, BN = FLASH_ROWS_BN, VSTR = FLASH_ROWS_VSTR, KSTR = HD + FLASH_ROWS_KPAD, NP = HD / 64;
  static_assert(BN == 64, "a fill is two 32-position steps: one per wave of a pair");
  // LDS, two buffers each: V^T tile [HD][VSTR] (filled transposed), K tile [BN][KSTR] (as the cache holds it)
  extern __shared__ __attribute__((aligned(16))) unsigned short fr_lds[];
  unsigned short* vt0 = fr_lds;
  unsigned short* kt0 = fr_lds + 2 * HD * VSTR;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int n = lane & 15, g = lane >> 4;
  const int head = blockIdx.y, j = head / (n_heads / n_kv);
  const int pos0 = *pos_d;
  // eight waves: waves w and w + 4 share 16 prompt rows and take alternate 32-position steps (two waves per SIMD: one's matrix work
  // runs under the other's softmax); their {m, l, O} states are merged once, at the end
  const int rw = wave & 3, par = wave >> 2;
  const int row_wg = blockIdx.x * 64, row_w = row_wg + rw * 16;
  const int row = row_w + n < B ? row_w + n : B - 1;  // (rows past the batch recompute the last row; never stored)
  // this wave's q rows as the B operand: lane (row n, dims 32 ks + 8 g .. + 8)
  f16x8 qb[KS];
#pragma unroll
  for (int ks = 0; ks < KS; ks++) {
    const f32x4* qp = (const f32x4*)(q + ((size_t)row * n_heads + head) * HD + 32 * ks + 8 * g);
    const f32x4 a = qp[0], b = qp[1];
    qb[ks] = f16x8{(_Float16)a[0], (_Float16)a[1], (_Float16)a[2], (_Float16)a[3], (_Float16)b[0], (_Float16)b[1], (_Float16)b[2], (_Float16)b[3]};
  }
  const unsigned short* kb = kc + (size_t)j * seq_cap * HD;
  const unsigned short* vb = vc + (size_t)j * seq_cap * HD;
  const int last_wg = pos0 + (row_wg + 63 < B ? row_wg + 63 : B - 1);  // last cached position any row of the workgroup sees
  const int last_w = pos0 + (row_w + 15 < B ? row_w + 15 : B - 1);     // ... any row of this wave
  const int my_last = pos0 + row;                                      // ... this lane's row
  const int ntiles = last_wg / BN + 1;
  f32x4 acc[DT];
#pragma unroll
  for (int dt = 0; dt < DT; dt++) acc[dt] = f32x4{0.f, 0.f, 0.f, 0.f};
  float m = -INFINITY, lsum = 0.f;
  // staging: thread t moves NP 16-byte pieces of the K tile and of the V tile per fill (piece = 8 dims of one position); a whole
  // tile of compute lies between the request and the commit
  i32x4 vreg[NP], kreg[NP];
  auto issue = [&](int tile) {
#pragma unroll
    for (int u = 0; u < NP; u++) {
      const int piece = tid + 512 * u, pr = piece / (HD / 8), pc = piece % (HD / 8);
      int pos = tile * BN + pr;
      pos = pos < seq_cap ? pos : seq_cap - 1;  // (past the cache only in its last tile; those positions are dead)
      kreg[u] = *(const i32x4*)(kb + (size_t)pos * HD + 8 * pc);
      vreg[u] = *(const i32x4*)(vb + (size_t)pos * HD + 8 * pc);
    }
  };
  auto commit = [&](int buf, int tile) {
    unsigned short* vt = vt0 + buf * HD * VSTR;
    unsigned short* kt = kt0 + buf * BN * KSTR;
#pragma unroll
    for (int u = 0; u < NP; u++) {
      const int piece = tid + 512 * u, pr = piece / (HD / 8), pc = piece % (HD / 8);
      // positions no row of the workgroup sees hold whatever the allocator or an earlier sequence left in the cache: their
      // probabilities are exactly 0, but 0 x NaN / Inf is not -- V enters the product as zeros there (K: the scores are masked)
      const bool live = tile * BN + pr <= last_wg;
      *(i32x4*)(kt + pr * KSTR + 8 * pc) = kreg[u];
#pragma unroll
      for (int i = 0; i < 4; i++) {
        const unsigned w = live ? (unsigned)vreg[u][i] : 0u;
        vt[(8 * pc + 2 * i) * VSTR + pr] = (unsigned short)(w & 0xffffu);
        vt[(8 * pc + 2 * i + 1) * VSTR + pr] = (unsigned short)(w >> 16);
      }
    }
  };
  issue(0);
  commit(0, 0);
  __syncthreads();
  for (int tile = 0; tile < ntiles; tile++) {
    const int buf = tile & 1;
    const unsigned short* vt = vt0 + buf * HD * VSTR;
    const unsigned short* kt = kt0 + buf * BN * KSTR;
    if (tile + 1 < ntiles) issue(tile + 1);
    {
      // 32 positions per step as two 16-position chunks a / b: two independent S^T chains, ONE agreement on the running maximum, and
      // the second product on 16x16x32 with k-slot (g, i) = position 4 g + i of chunk a, (g, 4 + i) = the same of chunk b -- for
      // both operands.  This wave's step of the tile: c2 = par.
      const int c2 = par;
      const int ca0 = tile * BN + 32 * c2, cb0 = ca0 + 16;
      if (ca0 <= last_w) {  // wave-uniform: some row of this wave sees the step
      f32x4 sa = {0.f, 0.f, 0.f, 0.f}, sb = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int ks = 0; ks < KS; ks++) {
        sa = __builtin_amdgcn_mfma_f32_16x16x32_f16(*(const f16x8*)(kt + (32 * c2 + n) * KSTR + 32 * ks + 8 * g), qb[ks], sa, 0, 0, 0);
        sb = __builtin_amdgcn_mfma_f32_16x16x32_f16(*(const f16x8*)(kt + (32 * c2 + 16 + n) * KSTR + 32 * ks + 8 * g), qb[ks], sb, 0, 0, 0);
      }
      float cm = -INFINITY;
#pragma unroll
      for (int i = 0; i < 4; i++) {  // causal mask (also: dead cache rows, whatever they hold)
        if (ca0 + 4 * g + i > my_last) sa[i] = -INFINITY;
        if (cb0 + 4 * g + i > my_last) sb[i] = -INFINITY;
        cm = fmaxf(cm, fmaxf(sa[i], sb[i]));
      }
      cm = fmaxf(cm, __shfl_xor(cm, 16));
      cm = fmaxf(cm, __shfl_xor(cm, 32));
      // (a row may have seen nothing yet -- the odd-step wave of a pair starts at position 32: m = mn = -inf, alpha = 1, p = 0)
      const float mn = fmaxf(m, cm);
      if (__any(mn > m)) {  // the running maximum of some row moved: rescale (rare after the first steps)
        const float alpha = mn == -INFINITY ? 1.0f : __expf(m - mn);
        lsum *= alpha;
#pragma unroll
        for (int dt = 0; dt < DT; dt++) acc[dt] = acc[dt] * alpha;
        m = mn;
      }
      f16x8 pT;
#pragma unroll
      for (int i = 0; i < 4; i++) {
        const _Float16 pa = m == -INFINITY ? (_Float16)0.0f : (_Float16)__expf(sa[i] - m);
        const _Float16 pb = m == -INFINITY ? (_Float16)0.0f : (_Float16)__expf(sb[i] - m);
        pT[i] = pa;
        pT[4 + i] = pb;
        lsum += (float)pa + (float)pb;  // the normalizer sums what the second product multiplies
      }
#pragma unroll
      for (int dt = 0; dt < DT; dt++) {
        const unsigned short* vr = vt + (16 * dt + n) * VSTR + 32 * c2 + 4 * g;
        const f16x4 va = *(const f16x4*)vr, vb2 = *(const f16x4*)(vr + 16);
        const f16x8 vf = {va[0], va[1], va[2], va[3], vb2[0], vb2[1], vb2[2], vb2[3]};
        acc[dt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(vf, pT, acc[dt], 0, 0, 0);
      }
      }
    }
    if (tile + 1 < ntiles) commit(buf ^ 1, tile + 1);  // those buffers were last read one tile ago (barrier below)
    __syncthreads();
  }
  lsum += __shfl_xor(lsum, 16);
  lsum += __shfl_xor(lsum, 32);
  // ---- the two waves of a pair meet: the odd-step wave hands {m, l, O} over through LDS (the tiles are done with: the loop's last
  // barrier is behind every wave), the even-step wave merges and stores
  // [lane][4 DT + 4] floats per row group: {O, m, l} padded to whole 16-byte pieces, so that the f32x4 accesses of every lane are
  // 16-byte aligned (a record of 4 DT + 2 floats put the odd lanes on 8-byte boundaries)
  float* xch = (float*)fr_lds + (size_t)rw * 64 * (4 * DT + 4);
  if (par == 1) {
    float* d = xch + lane * (4 * DT + 4);
#pragma unroll
    for (int dt = 0; dt < DT; dt++) *(f32x4*)(d + 4 * dt) = acc[dt];
    d[4 * DT] = m;
    d[4 * DT + 1] = lsum;
  }
  __syncthreads();
  if (par == 0 && row_w + n < B) {
    const float* d = xch + lane * (4 * DT + 4);
    const float m1 = d[4 * DT], l1 = d[4 * DT + 1];
    const float M = fmaxf(m, m1);  // (m is finite: the even-step wave saw position 0; m1 = -inf if the odd wave saw nothing)
    const float w0 = __expf(m - M), w1 = m1 == -INFINITY ? 0.0f : __expf(m1 - M);
    const float inv = 1.0f / (lsum * w0 + l1 * w1);
    float* o = out + ((size_t)(row_w + n) * n_heads + head) * HD;
#pragma unroll
    for (int dt = 0; dt < DT; dt++) *(f32x4*)(o + 16 * dt + 4 * g) = (acc[dt] * w0 + *(const f32x4*)(d + 4 * dt) * w1) * inv;
  }
}

// The same PV pass for R consecutive prompt rows per workgroup (batched prefill past 1024 positions): row r of the tile
// sees seq0 + r cached positions.  The V tile is fetched and transposed ONCE for the R rows x G heads -- every lane of the
// workgroup carries a chain (R * G * 16 = 256 for Llama-3's G = 4, R = 4) instead of 64 of 256, and V is read R times less
// often (a 4096-token prompt re-read V once per row: 340 of 582 ms).  Per (row, head, column) the arithmetic and its order
// are k_attn_pv's: bit-identical (test_long_prompt_attention_paths_are_bit_identical).
template <int G, int R>
__global__ __launch_bounds__(256) void k_attn_pv_rows(const unsigned short* __restrict__ p16, const unsigned short* __restrict__ vc,
                                                      const int* __restrict__ pos_d, float* __restrict__ out, int hd, int seq_cap,
                                                      int row0, int nrows) {
  static_assert(R * G * 16 <= 256, "one chain per lane");
  constexpr int T = ATTN_PV_TILE, ROW = ATTN_PV_ROW;
  __shared__ __attribute__((aligned(16))) unsigned vt[2][16 * ROW];
  __shared__ __attribute__((aligned(16))) unsigned pt[2][R * G * ROW];
  const int tid = threadIdx.x;
  const int nslice = hd / 32;
  const int j = blockIdx.x / nslice, sl = blockIdx.x % nslice;
  const int rt0 = (int)blockIdx.y * R;                       // first row of this tile within the launch
  const int rows_here = nrows - rt0 < R ? nrows - rt0 : R;   // >= 1
  const int seq0 = *pos_d + 1 + row0 + rt0;                  // cached positions of the tile's first row
  const int seq_max = seq0 + rows_here - 1;
  const size_t n_heads = (size_t)(gridDim.x / nslice) * G;
  p16 += (size_t)rt0 * n_heads * seq_cap;
  out += (size_t)(row0 + rt0) * n_heads * hd;
  const unsigned short* vbase = vc + (size_t)j * seq_cap * hd + sl * 32;
  const int ntiles = (seq_max + T - 1) / T;
  constexpr int PP = (R * G * (T / 8) + 255) / 256;  // probability pieces (8 positions of one (row, head)) per thread
  i32x4 vreg[4], preg[PP];
  auto issue = [&](int tile) {
    const int t0 = tile * T;
#pragma unroll
    for (int r = 0; r < 4; r++) {
      int t = t0 + (tid >> 2) + 64 * r;
      t = t < seq_cap ? t : seq_cap - 1;
      vreg[r] = *(const i32x4*)(vbase + (size_t)t * hd + (tid & 3) * 8);
    }
#pragma unroll
    for (int u = 0; u < PP; u++) {
      const int pc = tid + 256 * u;
      if (pc < R * G * (T / 8)) {
        const int rg = pc / (T / 8), c8 = pc % (T / 8), r = rg / G, g = rg % G;
        const int rr = r < rows_here ? r : rows_here - 1;  // rows past the batch: re-read the last row (never consumed)
        int t = t0 + c8 * 8;
        t = t + 8 <= seq_cap ? t : seq_cap - 8;
        preg[u] = *(const i32x4*)(p16 + ((size_t)rr * n_heads + (j * G + g)) * seq_cap + t);
      }
    }
  };
  auto commit = [&](int buf) {
#pragma unroll
    for (int r = 0; r < 4; r++) {
      const int tl = (tid >> 2) + 64 * r;
#pragma unroll
      for (int i = 0; i < 4; i++) vt[buf][((tid & 3) * 4 + i) * ROW + tl] = (unsigned)vreg[r][i];
    }
#pragma unroll
    for (int u = 0; u < PP; u++) {
      const int pc = tid + 256 * u;
      if (pc < R * G * (T / 8)) {
        const int rg = pc / (T / 8), c8 = pc % (T / 8);
        unsigned pp[8];
#pragma unroll
        for (int i = 0; i < 4; i++) {
          const unsigned w = (unsigned)preg[u][i];
          const unsigned a = w & 0xffffu, b = w >> 16;
          pp[2 * i] = a | (a << 16);
          pp[2 * i + 1] = b | (b << 16);
        }
        *(i32x4*)(&pt[buf][rg * ROW + c8 * 8]) = i32x4{(int)pp[0], (int)pp[1], (int)pp[2], (int)pp[3]};
        *(i32x4*)(&pt[buf][rg * ROW + c8 * 8 + 4]) = i32x4{(int)pp[4], (int)pp[5], (int)pp[6], (int)pp[7]};
      }
    }
  };
  // chain role: lane = (row r, head g, dim pair dp)
  const int rg = tid >> 4, dp = tid & 15, r = rg / G, g = rg % G;
  const bool chain = tid < R * G * 16 && r < rows_here;
  const int seq = seq0 + r;
  h16x2 c2 = {(_Float16)0.0f, (_Float16)0.0f};
  issue(0);
  commit(0);
  __syncthreads();
  for (int tile = 0; tile < ntiles; tile++) {
    const int buf = tile & 1;
    if (tile + 1 < ntiles) issue(tile + 1);
    if (chain && tile * T < seq) {
      const int nt = seq - tile * T < T ? seq - tile * T : T;
      const unsigned* vrow = &vt[buf][dp * ROW];
      const unsigned* prow = &pt[buf][rg * ROW];
      int t = 0;
#define PV_ROUND(NB)                                                                                         \
  for (; t + 8 * NB <= nt; t += 8 * NB) {                                                                    \
    i32x4 vq[2 * NB], pq[2 * NB];                                                                            \
    _Pragma("unroll") for (int b = 0; b < 2 * NB; b++) {                                                     \
      vq[b] = *(const i32x4*)(vrow + t + 4 * b);                                                             \
      pq[b] = *(const i32x4*)(prow + t + 4 * b);                                                             \
    }                                                                                                        \
    _Pragma("unroll") for (int b = 0; b < 2 * NB; b++) _Pragma("unroll") for (int u = 0; u < 4; u++) {       \
      const h16x2 pr = __builtin_bit_cast(h16x2, (unsigned)vq[b][u]) * __builtin_bit_cast(h16x2, (unsigned)pq[b][u]); \
      c2 = c2 + pr;                                                                                          \
    }                                                                                                        \
  }
      PV_ROUND(4)
      PV_ROUND(1)
#undef PV_ROUND
      for (; t < nt; t++) {
        const h16x2 pr = __builtin_bit_cast(h16x2, vrow[t]) * __builtin_bit_cast(h16x2, prow[t]);
        c2 = c2 + pr;
      }
    }
    if (tile + 1 < ntiles) commit(buf ^ 1);
    __syncthreads();
  }
  if (!chain) return;
  const int head = j * G + g;
  const int e0 = head * hd + sl * 32 + 2 * dp;
  float* o = out + (size_t)r * n_heads * hd;
  o[e0] = (float)c2[0];
  o[e0 + 1] = (float)c2[1];
}

// ---- batched-prefill attention: one workgroup = one kv head x R consecutive prompt rows x the G q heads of its
// group (Q = G * R queries).  Per (row, head) the arithmetic is k_attn's, value for value -- f32 dots in k order,
// softmax_row's table exp / sequential row sum (rows up to 1024 positions; longer prompts use k_attn) / true
// division, the f16 PV chain in position order -- but a K row is fetched once for the Q queries that score against
// it and a V element once for the Q / (256 / hd) chains a thread carries, instead of once per (row, head) workgroup:
// the per-row kernel moved 2.1 GB through L2 per layer for 512 prompt rows of the 8B shape.
template <bool KV16, int G, int R>
__global__ __launch_bounds__(256) void k_attn_tile(const float* __restrict__ q, const void* __restrict__ kc,
                                                   const void* __restrict__ vc, const int* __restrict__ pos_d,
                                                   const unsigned short* __restrict__ exp_tab, float* __restrict__ out,
                                                   int n_heads, int n_kv, int hd, int seq_cap, int n_rows, int sstride) {
  constexpr int Q = G * R;
  extern __shared__ float lds[];
  float* qs = lds;            // [Q][hd]: q rows (rounded to f16 for the f16 cache, batch_matmul.rs:39)
  float* sc = lds + Q * hd;   // [Q][sstride]: scores, then probabilities
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int kvh = blockIdx.x, r0 = blockIdx.y * R;
  const int pos0 = *pos_d;
  const int dim = n_heads * hd;
  auto head_of = [&](int j) { return KV16 ? kvh * G + j : kvh + j * n_kv; };  // batch_matmul.rs:61-67 GQA maps
  const int rows_here = n_rows - r0 < R ? n_rows - r0 : R;
  for (int e = tid; e < Q * hd; e += 256) {  // query qi = r * G + j
    const int qi = e / hd, i = e - qi * hd, r = qi / G, j = qi - r * G;
    const float v = r < rows_here ? q[(size_t)(r0 + r) * dim + head_of(j) * hd + i] : 0.0f;
    if (KV16)
      ((unsigned short*)qs)[e] = f2h(v);  // f16 cache: q staged as f16, the dots run on v_fma_mix_f32 (exact: see k_attn_s)
    else
      qs[e] = v;
  }
  __syncthreads();
  // ---- scores + softmax: wave w owns the QW = Q / 4 queries w * QW .. (one prompt row: its causal length bounds the
  // loop), lane = cached position; a K row is fetched once per wave and scored against the wave's queries
  constexpr int QW = Q / 4;
  static_assert(Q % 4 == 0 && (G % QW == 0 || QW % G == 0), "a wave's queries belong to one row");
  {
    const int q0 = wave * QW, rw = q0 / G;
    if (rw < rows_here) {
      const int seq = pos0 + r0 + rw + 1;
      for (int t = lane; t < seq; t += 64) {
        float acc[QW];
#pragma unroll
        for (int u = 0; u < QW; u++) acc[u] = 0.0f;
        if (KV16) {
          const unsigned short* kr = (const unsigned short*)kc + ((size_t)kvh * seq_cap + t) * hd;
          typedef _Float16 h2q __attribute__((ext_vector_type(2)));
          const unsigned short* q16 = (const unsigned short*)qs;
          for (int i = 0; i < hd; i += 16) {  // hd % 16 == 0 (host check); products added in k order per query
            const i32x4 k0 = *(const i32x4*)(kr + i), k1 = *(const i32x4*)(kr + i + 8);
#pragma unroll
            for (int u = 0; u < QW; u++) {
              const i32x4 qa = *(const i32x4*)(q16 + (q0 + u) * hd + i), qb = *(const i32x4*)(q16 + (q0 + u) * hd + i + 8);
#pragma unroll
              for (int w4 = 0; w4 < 4; w4++) {
                const h2q kh = __builtin_bit_cast(h2q, (unsigned)k0[w4]), qh = __builtin_bit_cast(h2q, (unsigned)qa[w4]);
                acc[u] = __builtin_fmaf((float)qh[0], (float)kh[0], acc[u]);
                acc[u] = __builtin_fmaf((float)qh[1], (float)kh[1], acc[u]);
              }
#pragma unroll
              for (int w4 = 0; w4 < 4; w4++) {
                const h2q kh = __builtin_bit_cast(h2q, (unsigned)k1[w4]), qh = __builtin_bit_cast(h2q, (unsigned)qb[w4]);
                acc[u] = __builtin_fmaf((float)qh[0], (float)kh[0], acc[u]);
                acc[u] = __builtin_fmaf((float)qh[1], (float)kh[1], acc[u]);
              }
            }
          }
        } else {
          const float* kr = (const float*)kc + ((size_t)kvh * seq_cap + t) * hd;
          for (int i = 0; i < hd; i += 4) {
            const f32x4 kv = *(const f32x4*)(kr + i);
#pragma unroll
            for (int u = 0; u < QW; u++) {
              const f32x4 qv = *(const f32x4*)(qs + (q0 + u) * hd + i);
              acc[u] += qv[0] * kv[0];
              acc[u] += qv[1] * kv[1];
              acc[u] += qv[2] * kv[2];
              acc[u] += qv[3] * kv[3];
            }
          }
        }
#pragma unroll
        for (int u = 0; u < QW; u++) sc[(q0 + u) * sstride + t] = acc[u];
      }
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
      __builtin_amdgcn_wave_barrier();
      // softmax per query (softmax.rs:36-54; softmax_row with the <= 1024 sequential row sum), by the same wave; the
      // QW sequential row sums (one dependent v_add chain each) run interleaved
#pragma unroll
      for (int u = 0; u < QW; u++) {
        float* srow = sc + (q0 + u) * sstride;
        float mx = -INFINITY;
        for (int t = lane; t < seq; t += 64) mx = fmaxf(mx, srow[t]);
        mx = wave_max_f32(mx);
        for (int t = lane; t < seq; t += 64) srow[t] = exp_cached_f(srow[t] - mx, exp_tab);
      }
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
      __builtin_amdgcn_wave_barrier();
      float sum[QW];
#pragma unroll
      for (int u = 0; u < QW; u++) sum[u] = 0.0f;
      for (int base = 0; base < seq; base += 64) {
        float v[QW];
#pragma unroll
        for (int u = 0; u < QW; u++) v[u] = base + lane < seq ? sc[(q0 + u) * sstride + base + lane] : 0.0f;
#pragma unroll
        for (int i = 0; i < 64; i++)
#pragma unroll
          for (int u = 0; u < QW; u++) sum[u] += rl_f(v[u], i);  // lanes past `seq` add +0.0 (exact)
      }
#pragma unroll
      for (int u = 0; u < QW; u++) {
        float* srow = sc + (q0 + u) * sstride;
        for (int t = lane; t < seq; t += 64) {
          const float pv = srow[t] / sum[u];
          if (KV16) {  // quantize_f32_f16 of the lhs (batch_matmul.rs:39), stored as the pair {p, p} the PV chains multiply by
            const unsigned h = (unsigned)f2h(pv);
            ((unsigned*)srow)[t] = h | (h << 16);
          } else {
            srow[t] = pv;
          }
        }
      }
    }
  }
  __syncthreads();
  // ---- out[qi][n] = sum_t p[qi][t] * V[t][n].  f16 cache: thread = (pair of columns, query group), the chains run on
  // v_pk_mul_f16 / v_pk_add_f16 (per half exactly the scalar product-round, sum-round of buf_f16.rs:152-163); a V
  // pair feeds every chain the thread carries.  f32 cache: thread = (column, query group), plain f32.
  constexpr int QS = Q / 2 > 0 ? Q / 2 : 1;  // chains per thread (>= 2 query groups); unused slots have lim = 0
  const int tpg = KV16 ? hd / 2 : hd;      // threads per query group
  const int ngrp = 256 / tpg;
  const int n = tid % tpg, grp = tid / tpg;
  if (grp >= ngrp) return;  // ngrp >= 2 (host check), so Q / 2 chain slots cover the Q queries
  const int ch = (Q + ngrp - 1) / ngrp;  // consecutive queries per group: normally the heads of ONE row (same length)
  int lim[QS];
  const float* prow[QS];
  int lim_lo = 0x7fffffff, lim_hi = 0;
#pragma unroll
  for (int s2 = 0; s2 < QS; s2++) {
    const int qi = grp * ch + s2;
    const bool live = s2 < ch && qi < Q && qi / G < rows_here;
    lim[s2] = live ? pos0 + r0 + qi / G + 1 : 0;
    prow[s2] = sc + (live ? qi : 0) * sstride;
    if (live) {
      lim_lo = lim[s2] < lim_lo ? lim[s2] : lim_lo;
      lim_hi = lim[s2] > lim_hi ? lim[s2] : lim_hi;
    }
  }
  if (lim_hi == 0) return;
  if (KV16) {
    typedef _Float16 h2v __attribute__((ext_vector_type(2)));
    const unsigned* vr = (const unsigned*)((const unsigned short*)vc + (size_t)kvh * seq_cap * hd) + n;
    const int vs = hd / 2;  // dwords per V row
    h2v c[QS];
#pragma unroll
    for (int s2 = 0; s2 < QS; s2++) c[s2] = h2v{(_Float16)0.0f, (_Float16)0.0f};
    // the common case: every chain of the thread has the same causal length (one row) and QS / 2 live chains
    const bool uniform = lim_lo == lim_hi;
    int t0 = 0;
    if (uniform) {
      for (; t0 + 4 <= lim_hi; t0 += 4) {
        unsigned vv[4];
#pragma unroll
        for (int u = 0; u < 4; u++) vv[u] = vr[(size_t)(t0 + u) * vs];
#pragma unroll
        for (int s2 = 0; s2 < QS; s2++) {
          if (lim[s2]) {  // thread-constant
            // four {p, p} pairs, read as scalars (element extraction from a freshly loaded ext-vector feeding
            // bit_casts was miscompiled here: every element became element 0)
            const unsigned* pq = (const unsigned*)prow[s2] + t0;
            unsigned pp[4];
#pragma unroll
            for (int u = 0; u < 4; u++) pp[u] = pq[u];
#pragma unroll
            for (int u = 0; u < 4; u++) {
              const h2v prod = __builtin_bit_cast(h2v, vv[u]) * __builtin_bit_cast(h2v, pp[u]);
              c[s2] = c[s2] + prod;
            }
          }
        }
      }
    }
    for (; t0 < lim_hi; t0++) {  // tail / mixed lengths
      const h2v vp = __builtin_bit_cast(h2v, vr[(size_t)t0 * vs]);
#pragma unroll
      for (int s2 = 0; s2 < QS; s2++) {
        if (t0 < lim[s2]) {
          const h2v prod = vp * __builtin_bit_cast(h2v, ((const unsigned*)prow[s2])[t0]);
          c[s2] = c[s2] + prod;
        }
      }
    }
#pragma unroll
    for (int s2 = 0; s2 < QS; s2++) {
      const int qi = grp * ch + s2;
      if (lim[s2]) {
        float* o = out + (size_t)(r0 + qi / G) * dim + head_of(qi % G) * hd + 2 * n;
        o[0] = (float)c[s2][0];
        o[1] = (float)c[s2][1];
      }
    }
  } else {
    const float* vr = (const float*)vc + (size_t)kvh * seq_cap * hd + n;
    float c[QS];
#pragma unroll
    for (int s2 = 0; s2 < QS; s2++) c[s2] = 0.0f;
    for (int t0 = 0; t0 < lim_hi; t0 += 8) {
      float vv[8];
#pragma unroll
      for (int u = 0; u < 8; u++) vv[u] = t0 + u < lim_hi ? vr[(size_t)(t0 + u) * hd] : 0.0f;
#pragma unroll
      for (int u = 0; u < 8; u++) {
        const int t = t0 + u;
#pragma unroll
        for (int s2 = 0; s2 < QS; s2++) {
          if (t < lim[s2]) c[s2] += prow[s2][t] * vv[u];
        }
      }
    }
#pragma unroll
    for (int s2 = 0; s2 < QS; s2++) {
      const int qi = grp * ch + s2;
      if (lim[s2]) out[(size_t)(r0 + qi / G) * dim + head_of(qi % G) * hd + n] = c[s2];
    }
  }
}

}  // namespace crabml_hip

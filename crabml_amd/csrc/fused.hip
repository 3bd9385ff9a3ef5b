// fused.hip -- the fused Llama decode step (crabml_hip_llama_*): the hot path as 8 kernels per layer,
// replayed from one hipGraph with token id / position resident in device memory.
//
// It serves exactly the op sequence Llama2Runner<T> issues for one token (crabml-llama2/src/llama2.rs):
//   forward_llama :213-281, forward_multi_query_attention :527-603, forward_ffn :605-638, classifier :184-211
// with the reference's arithmetic (each fused stage cites the primitive it folds in).  Why fuse: the per-op
// trait path is launch-bound (31 launches/layer, GPU busy 1/3 of the time; profiles/r01_trait_path_kernel_
// trace.md).  The GEMV stages use the same lane-per-block / R-rows-per-wave mapping as gemv.hip and stay
// HBM-bound; the small stages are folded into their producers/consumers so activations never round-trip
// through extra launches:
//   k_norm_quant   rms_norm_inplace + mul_inplace(weight) + quantize_f32_q8_0        (1 workgroup)
//   k_qkv          wq/wk/wv matmul_vec + rope_inplace(q,k) + scale_inplace(q) + concatenate(k,v -> KV cache)
//   k_attn         batch_matmul(q,K^T) + softmax_inplace + batch_matmul(p,V) [+ quantize for wo]
//   k_gemv_res     wo / ffn_down matmul_vec + add_inplace(residual)
//   k_gateup       ffn_gate/ffn_up matmul_vec + silu_inplace + mul_inplace
//   k_argmax_step  greedy sampler (last maximum) + token/position advance
#include <cmath>

#include "dequant.hpp"
#include "gemv_core.hpp"
#include "kernels.hpp"

namespace crabml_hip {

// exp_f32_cached (buf_f32.rs:29-35)
__device__ __forceinline__ float exp_cached_f(float x, const unsigned short* __restrict__ table) {
  return h2f(table[f2h(x)]);
}

// ---- the rhs quantizer of matmul_vec, one 32-lane half-wave per 32-element block ------------------------------
// Q81 = false: Q8_0 (buf_q8_0.rs:87-134: d = max|x| / 127, q = trunc(x / d) with the simd cast's NaN -> 0; aux = the
// i32 sum of the block's quants -- exact, derived, used for Q4_0's -8 offset).  Q81 = true: Q8_1 (buf_q8_1.rs:90-129:
// q = trunc(clamp(x / d, -128, 127)) with NaN -> -128, aux = the f16 s = d * sum q).  All 32 lanes of the half-wave
// call it (dead lanes with live = false and v = 0).
struct QLane {
  signed char q;
  unsigned short d;
  int aux;
};
template <bool Q81>
__device__ __forceinline__ QLane quant_lane32(float v, bool live) {
  QLane o;
  const float amax = half_max_f32(fabsf(v));
  const float dd = amax / 127.0f;
  o.d = f2h(dd);
  if constexpr (!Q81) {
    const int qi = rs_f32_as_i32(v / dd);
    o.q = (signed char)(unsigned char)((unsigned)qi & 0xffu);  // `as i8` from i32 wraps
    o.aux = half_sum_i32(live ? (int)o.q : 0);
  } else {
    const float c = fminf(fmaxf(v / dd, -128.0f), 127.0f);  // Rust f32::max / min return the non-NaN operand
    const int qi = (int)c;
    o.q = (signed char)qi;
    const int s = half_sum_i32(live ? qi : 0);
    o.aux = (int)f2h((float)s * dd);
  }
  return o;
}
template <bool Q81>
__device__ __forceinline__ void store_qaux(void* aux, int blk, int v) {
  if constexpr (Q81)
    ((unsigned short*)aux)[blk] = (unsigned short)v;
  else
    ((int*)aux)[blk] = v;
}

// ---- weight prefetch into the Infinity Cache ---------------------------------------------------------
// The norm+quantize and attention stages are latency-bound single-/few-workgroup kernels: HBM idles for
// ~6-8 us while they run.  Spare workgroups of those launches (one per otherwise idle CU) stream the NEXT
// GEMV's weights with plain loads and drop them: the lines land in the 256 MiB memory-side Infinity Cache,
// so the following HBM-bound GEMV starts on warm data.  Pure performance hint: no result depends on it.
struct PrefetchPlan {
  const void* p[3];
  unsigned long long n[3];  // bytes (multiples of 16)
  int* sink;
};
__device__ __forceinline__ void prefetch_wg(const PrefetchPlan& pf, int wg, int nwg) {
  int acc = 0;
#pragma unroll 1
  for (int sp = 0; sp < 3; sp++) {
    const i32x4* base = (const i32x4*)pf.p[sp];
    const size_t n16 = pf.n[sp] / 16;
    if (!base || n16 == 0) continue;
    const size_t per = (n16 + nwg - 1) / nwg;
    const size_t lo = (size_t)wg * per, hi = lo + per < n16 ? lo + per : n16;
    size_t i = lo + threadIdx.x;
    const size_t st = blockDim.x;
    for (; i + 3 * st < hi; i += 4 * st) {
      i32x4 a = base[i], b = base[i + st], c = base[i + 2 * st], d = base[i + 3 * st];
      acc ^= a[0] ^ b[1] ^ c[2] ^ d[3];
    }
    for (; i < hi; i += st) acc ^= base[i][0];
  }
  if (acc == 0x7eadbeef) *pf.sink = acc;  // never true in practice; keeps the loads alive
}

// ---- embedding lookup: copy_rows_from(token_embed, [token]) (llama2.rs:222-223) ----------------------
__global__ __launch_bounds__(256) void k_embed(const char* __restrict__ w, int dtype, size_t off_scale,
                                               const int* __restrict__ token_d, int dim, float* __restrict__ x) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= dim) return;
  // blockIdx.y: row of a prefill batch (token ids and output rows are consecutive); 0 for a decode step
  x[(size_t)blockIdx.y * dim + i] = dequant_elem(w, dtype, off_scale, (size_t)token_d[blockIdx.y] * dim + i);
}

// ---- rmsnorm * weight -> Q8_0 planes ------------------------------------------------------------------
// rms_norm.rs:33-46 (ordered 32-chunk sums, serial chunk accumulation, true division), arithmetic.rs:57-66
// (x * w), buf_q8_0.rs:87-134 (truncating quantizer).  x itself is left untouched: it is the residual.
// Executed by ONE 1024-thread workgroup (16 waves: 4 per SIMD, so the two IEEE divisions per element
// overlap across waves).  It is pure latency, so every global load (x and the norm weight) is issued up
// front in one batch and kept in registers (NIT values per thread); the ordered chunk sums are taken from
// an LDS copy.  Outputs (q / d / isum) may live in LDS (GEMV prologue) or in global memory.
struct NormLds {  // carved from dynamic LDS: xs[cols] f32 | chunk_sums[cols/32] f32
  float* xs;
  float* chunk_sums;
};
__host__ __device__ inline size_t norm_lds_bytes(int cols) { return (size_t)(cols + cols / 32) * sizeof(float); }

// QUANT = false: the normalized row goes to xn_out as f32 (formats whose rhs is not Q8_0 quantize it afterwards)
template <int NIT, bool QUANT, bool Q81 = false>  // cols <= NIT * 1024, blockDim.x == 1024; ends with the outputs written
__device__ __forceinline__ void norm_quant_block(float* __restrict__ x, const float* __restrict__ addv,
                                                 const float* __restrict__ w, int cols, float eps, NormLds L,
                                                 float* s_rms, signed char* q, unsigned short* d, void* isum,
                                                 float* __restrict__ xn_out, int half) {
  const int nchunks = cols / 32;
  const int tid = threadIdx.x;
  float xv[NIT], wv[NIT];
#pragma unroll
  for (int it = 0; it < NIT; it++) {
    int i = it * 1024 + tid;
    xv[it] = i < cols ? x[i] : 0.f;
    wv[it] = i < cols ? w[i] : 0.f;
    // tensor-parallel: the all-reduced wo / ffn_down output is added to the residual stream here
    // (x = matmul_out + x, llama2.rs:266 / :636) and written back
    if (addv != nullptr && i < cols) {
      xv[it] = addv[i] + xv[it];
      x[i] = xv[it];
    }
  }
#pragma unroll
  for (int it = 0; it < NIT; it++) {
    int i = it * 1024 + tid;
    if (i < cols) L.xs[i] = xv[it];
  }
  __syncthreads();
  for (int c = tid; c < nchunks; c += 1024) {
    const f32x4* p = (const f32x4*)(L.xs + c * 32);
    float s = -0.0f, s1 = -0.0f;
#pragma unroll
    for (int j = 0; j < 8; j++) {
      f32x4 t = p[j];
      float& a = (half && j >= 4) ? s1 : s;
      a += t[0] * t[0];
      a += t[1] * t[1];
      a += t[2] * t[2];
      a += t[3] * t[3];
    }
    // half (fast mode): chunk = (rows 0..15 in order) + (rows 16..31 in order), the split the wo / ffn_down norm
    // epilogue uses (two workgroups per chunk); otherwise the reference's 32-element scan (rms_norm.rs:35-38)
    L.chunk_sums[c] = half ? s + s1 : s;
  }
  __syncthreads();
  if (tid < 64) {
    // chunk sums are added strictly in chunk order (rms_norm.rs:35-40): wave 0 holds them in registers and
    // v_readlane feeds a single dependent v_add chain.  Lanes past nchunks contribute +0.0 (exact).
    float sum = 0.0f;
    for (int base = 0; base < nchunks; base += 64) {
      float v = base + tid < nchunks ? L.chunk_sums[base + tid] : 0.0f;
#pragma unroll
      for (int i = 0; i < 64; i++) sum += rl_f(v, i);
    }
    if (tid == 0) *s_rms = sqrtf(sum / (float)cols + eps);
  }
  __syncthreads();
  const float rms = *s_rms;
#pragma unroll
  for (int it = 0; it < NIT; it++) {
    int i = it * 1024 + tid;
    if (it * 1024 < cols) {  // wave-uniform; 32-lane halves are entirely in or out of range (cols % 32 == 0)
      bool live = i < cols;
      float v = live ? (xv[it] / rms) * wv[it] : 0.f;
      if constexpr (!QUANT) {
        if (live) xn_out[i] = v;
        continue;
      }
      const QLane o = quant_lane32<Q81>(v, live);
      if (live) {
        q[i] = o.q;
        if ((tid & 31) == 0) {
          d[i >> 5] = o.d;
          store_qaux<Q81>(isum, i >> 5, o.aux);
        }
      }
    }
  }
}

template <int NIT, bool Q81>
__global__ __launch_bounds__(1024) void k_norm_quant(float* __restrict__ x, const float* __restrict__ addv,
                                                    const float* __restrict__ w, int cols, float eps,
                                                    signed char* __restrict__ q, unsigned short* __restrict__ d,
                                                    void* __restrict__ isum, PrefetchPlan pf, int half) {
  if (blockIdx.x > 0) {  // spare workgroups: warm the Infinity Cache with the next GEMV's weights
    prefetch_wg(pf, blockIdx.x - 1, gridDim.x - 1);
    return;
  }
  extern __shared__ float lds[];
  __shared__ float s_rms;
  NormLds L{lds, lds + cols};
  norm_quant_block<NIT, true, Q81>(x, addv, w, cols, eps, L, &s_rms, q, d, isum, nullptr, half);
}
template <int NIT>
__global__ __launch_bounds__(1024) void k_norm_f32(float* __restrict__ x, const float* __restrict__ addv,
                                                  const float* __restrict__ w, int cols, float eps, float* __restrict__ xn, int half) {
  extern __shared__ float lds[];
  __shared__ float s_rms;
  NormLds L{lds, lds + cols};
  norm_quant_block<NIT, false>(x, addv, w, cols, eps, L, &s_rms, nullptr, nullptr, nullptr, xn, half);
}

// batched prefill: one workgroup per row of x (rows, cols) -> xn (rows, cols)
template <int NIT>
__global__ __launch_bounds__(1024) void k_norm_f32_rows(float* __restrict__ x, const float* __restrict__ w, int cols, float eps,
                                                       float* __restrict__ xn, int half) {
  extern __shared__ float lds[];
  __shared__ float s_rms;
  NormLds L{lds, lds + cols};
  norm_quant_block<NIT, false>(x + (size_t)blockIdx.x * cols, nullptr, w, cols, eps, L, &s_rms, nullptr, nullptr, nullptr,
                               xn + (size_t)blockIdx.x * cols, half);
}

// ---- QKV epilogue: rope (rope.rs:47-63) + q scale (llama2.rs:565) + KV append (concatenate.rs:172-204) ---
struct QkvEpi {
  float* q_out;       // (n_heads * hd) f32, roped and scaled
  void* kc;           // K cache of this layer [n_kv][seq_cap][hd]
  void* vc;
  const float* rope;  // [seq_cap][npairs][2] (cos, sin)
  const int* pos_d;
  float scale;        // 1 / sqrt(hd)
  int dim, kv_dim, hd, rope_dim, npairs, seq_cap, kv16;
};

// position + rotation for the pair starting at row0, loaded early (before the weight stream is consumed)
struct QkvPre {
  int pos;
  float c, s;
  bool rot;
};
__device__ __forceinline__ QkvPre qkv_preload(const QkvEpi& e, int row0, int row_of_batch = 0) {
  QkvPre p;
  p.pos = *e.pos_d + row_of_batch;
  p.c = 1.f;
  p.s = 0.f;
  p.rot = false;
  if (row0 < e.dim + e.kv_dim) {
    const int i = (row0 < e.dim ? row0 : row0 - e.dim) % e.hd;
    if (i < e.rope_dim) {
      const float* cs = e.rope + ((size_t)p.pos * e.npairs + (i >> 1)) * 2;
      p.c = cs[0];
      p.s = cs[1];
      p.rot = true;
    }
  }
  return p;
}
__device__ __forceinline__ void qkv_epilogue(const QkvEpi& e, const QkvPre& pre, int row0, float s0, float s1) {
  const int pos = pre.pos;
  if (row0 < e.dim + e.kv_dim) {  // q or k: rotate the (even, odd) pair
    const int i = (row0 < e.dim ? row0 : row0 - e.dim) % e.hd;
    float r0 = s0, r1 = s1;
    if (pre.rot) {
      float c = pre.c, s = pre.s;
      r0 = s0 * c - s1 * s;
      r1 = s0 * s + s1 * c;
    }
    if (row0 < e.dim) {
      e.q_out[row0] = r0 * e.scale;
      e.q_out[row0 + 1] = r1 * e.scale;
    } else {
      const int kr = row0 - e.dim;
      const size_t o = ((size_t)(kr / e.hd) * e.seq_cap + pos) * e.hd + i;
      if (e.kv16) {
        ((unsigned short*)e.kc)[o] = f2h(r0);
        ((unsigned short*)e.kc)[o + 1] = f2h(r1);
      } else {
        ((float*)e.kc)[o] = r0;
        ((float*)e.kc)[o + 1] = r1;
      }
    }
  } else {
    const int vr = row0 - e.dim - e.kv_dim;
    const size_t o = ((size_t)(vr / e.hd) * e.seq_cap + pos) * e.hd + (vr % e.hd);
    if (e.kv16) {
      ((unsigned short*)e.vc)[o] = f2h(s0);
      ((unsigned short*)e.vc)[o + 1] = f2h(s1);
    } else {
      ((float*)e.vc)[o] = s0;
      ((float*)e.vc)[o + 1] = s1;
    }
  }
}

struct Planes {
  const i32x4* q;
  const unsigned short* d;
};
// a Q6_K matrix standing in for one of a Q4_K layer's (llama.cpp *_K_M mixes): base = nullptr means "not used"
struct Planes6 {
  const char* base;
  size_t off_qh;
};

template <int FMT>
__global__ __launch_bounds__(128) void k_qkv(Planes wq, Planes wk, Planes wv, typename ActOf<FMT>::type act, int nb, QkvEpi e,
                                             Planes6 wv6) {
  const int lane = threadIdx.x & 63;
  const int wave = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
  const int row0 = wave * 2;
  const int total = e.dim + 2 * e.kv_dim;
  if (row0 >= total) return;
  Planes w;
  int local, m;
  if (row0 < e.dim) {
    w = wq; local = row0; m = e.dim;
  } else if (row0 < e.dim + e.kv_dim) {
    w = wk; local = row0 - e.dim; m = e.kv_dim;
  } else {
    w = wv; local = row0 - e.dim - e.kv_dim; m = e.kv_dim;
  }
  QkvPre pre{};
  if (lane == 0) pre = qkv_preload(e, row0);
  float acc[2];
  bool done = false;
  if constexpr (FMT == CRABML_HIP_Q4_K) {
    if (wv6.base != nullptr && row0 >= e.dim + e.kv_dim) {  // the V rows of this layer are Q6_K (wave-uniform)
      rows_partial_q6k<2>(wv6.base, wv6.off_qh, act, local, m, nb, lane, acc);
      done = true;
    }
  }
  if (!done) rows_dot<FMT, 2>(w.q, w.d, act, local, m, nb, lane, acc);
  float s0 = wave_sum_f32(acc[0]), s1 = wave_sum_f32(acc[1]);
  if (lane == 0) qkv_epilogue(e, pre, row0, s0, s1);
}
// strict mode: the three GEMVs ran in scalar order into tmp[dim + 2 kv_dim]; apply the same epilogue
__global__ __launch_bounds__(256) void k_qkv_epi(const float* __restrict__ tmp, QkvEpi e) {
  int p = blockIdx.x * blockDim.x + threadIdx.x;
  int total = (e.dim + 2 * e.kv_dim) / 2;
  if (p < total) qkv_epilogue(e, qkv_preload(e, 2 * p), 2 * p, tmp[2 * p], tmp[2 * p + 1]);
}

// batched prefill: the three GEMMs wrote qb (B, dim), kb / vb (B, kv_dim); row r is position *pos_d + r
__global__ __launch_bounds__(256) void k_qkv_epi_rows(const float* __restrict__ qb, const float* __restrict__ kb,
                                                     const float* __restrict__ vb, QkvEpi e) {
  const int p = blockIdx.x * blockDim.x + threadIdx.x, r = blockIdx.y;
  if (p >= (e.dim + 2 * e.kv_dim) / 2) return;
  const int row0 = 2 * p;
  const float* src = row0 < e.dim              ? qb + (size_t)r * e.dim + row0
                     : row0 < e.dim + e.kv_dim ? kb + (size_t)r * e.kv_dim + (row0 - e.dim)
                                               : vb + (size_t)r * e.kv_dim + (row0 - e.dim - e.kv_dim);
  QkvEpi er = e;
  er.q_out = e.q_out + (size_t)r * e.dim;
  qkv_epilogue(er, qkv_preload(e, row0, r), row0, src[0], src[1]);
}

// softmax.rs:36-54 over scores[0..seq) in LDS, in place, by a 256-thread workgroup: max, exp through the f16 table,
// row sum sequential up to 1024 positions (bit-exact) and a block tree beyond, true division.  F16: the
// probabilities are then rounded to f16 (quantize_f32_f16 of the lhs, batch_matmul.rs:39).  Ends with a barrier.
template <bool F16>
__device__ __forceinline__ void softmax_row(float* scores, int seq, const unsigned short* __restrict__ exp_tab, float* s_red,
                                            float* s_val_p) {
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  float mx = -INFINITY;
  for (int t = tid; t < seq; t += blockDim.x) mx = fmaxf(mx, scores[t]);
  mx = wave_max_f32(mx);
  if (lane == 0) s_red[wave] = mx;
  __syncthreads();
  mx = fmaxf(fmaxf(s_red[0], s_red[1]), fmaxf(s_red[2], s_red[3]));
  __syncthreads();
  float part = 0.0f;
  {
    // the table lookups are independent global gathers: 8 in flight per thread (long rows), summed in t order
    const int bd = blockDim.x;
    int t = tid;
    for (; t + 7 * bd < seq; t += 8 * bd) {
      float ev[8];
#pragma unroll
      for (int u = 0; u < 8; u++) ev[u] = exp_cached_f(scores[t + u * bd] - mx, exp_tab);
#pragma unroll
      for (int u = 0; u < 8; u++) {
        scores[t + u * bd] = ev[u];
        part += ev[u];
      }
    }
    for (; t < seq; t += bd) {
      float ev = exp_cached_f(scores[t] - mx, exp_tab);
      scores[t] = ev;
      part += ev;
    }
  }
  __syncthreads();
  if (seq <= 1024) {
    if (tid < 64) {
      // sequential row sum (softmax.rs:43-48) without an LDS round trip per add: wave 0 holds 64 values per
      // pass in registers and v_readlane feeds one dependent v_add chain; lanes past `seq` add +0.0 (exact)
      float sum = 0.0f;
      for (int base = 0; base < seq; base += 64) {
        float v = base + tid < seq ? scores[base + tid] : 0.0f;
#pragma unroll
        for (int i = 0; i < 64; i++) sum += rl_f(v, i);
      }
      if (tid == 0) *s_val_p = sum;
    }
  } else {
    part = wave_sum_f32(part);
    if (lane == 0) s_red[wave] = part;
    __syncthreads();
    if (tid == 0) *s_val_p = (s_red[0] + s_red[1]) + (s_red[2] + s_red[3]);
  }
  __syncthreads();
  const float sum = *s_val_p;
  for (int t = tid; t < seq; t += blockDim.x) {
    float pv = scores[t] / sum;
    scores[t] = F16 ? h2f(f2h(pv)) : pv;  // quantize_f32_f16 of the lhs (batch_matmul.rs:39), done once
  }
  __syncthreads();
}

// ---- attention: one workgroup per head -------------------------------------------------------------------
// batch_matmul.rs: f16 cache -> q rounded to f16, f32-accumulated QK^T in k order (buf_f16.rs:83-97),
// GQA head = h / (n_heads/n_kv); PV accumulated in f16 with a rounding after the product and after the sum
// (buf_f16.rs:152-163).  f32 cache -> plain f32 loops, kv head = h % n_kv (batch_matmul.rs:61-67).
// softmax.rs:36-54 with the f16 exp table; the row sum is sequential (bit-exact) up to 1024 positions and a
// block tree beyond that (documented tolerance 1e-6 relative).
template <bool KV16>
__global__ __launch_bounds__(256) void k_attn(const float* __restrict__ q, const void* __restrict__ kc,
                                              const void* __restrict__ vc, const int* __restrict__ pos_d,
                                              const unsigned short* __restrict__ exp_tab, float* __restrict__ out,
                                              signed char* __restrict__ xq, unsigned short* __restrict__ xd,
                                              void* __restrict__ xisum, int n_heads, int n_kv, int hd, int seq_cap,
                                              PrefetchPlan pf, int q81) {
  if ((int)blockIdx.x >= n_heads) {
    prefetch_wg(pf, blockIdx.x - n_heads, gridDim.x - n_heads);
    return;
  }
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
  // ---- softmax (in place; probabilities rounded to f16 for the f16 cache)
  softmax_row<KV16>(scores, seq, exp_tab, s_red, &s_val);
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
                                                     int n_kv, int hd, int seq_cap, int nsplit) {
  // one thread per (cached position, q head of the group): the G threads of a position sit in adjacent lanes and
  // read the same K row (one fetch); each runs its own k-ordered f32 accumulation (buf_f16.rs:83-97)
  extern __shared__ float lds[];  // qs[G][hd]
  constexpr int TS = 256 / G;     // positions per workgroup
  const int tid = threadIdx.x;
  const int j = blockIdx.x / nsplit, sp = blockIdx.x % nsplit;
  const int seq = *pos_d + 1;
  if (sp * TS >= seq) return;
  for (int idx = tid; idx < G * hd; idx += 256) {
    const int g = idx / hd, i = idx - g * hd;
    lds[idx] = h2f(f2h(q[(size_t)(j * G + g) * hd + i]));  // quantize_f32_f16(bufa) (batch_matmul.rs:39)
  }
  __syncthreads();
  const int g = tid % G;
  const int t = sp * TS + tid / G;
  if (t >= seq) return;
  const unsigned short* kr = kc + ((size_t)j * seq_cap + t) * hd;
  const float* qg = lds + g * hd;
  float acc = 0.0f;
  int i = 0;
  for (; i + 64 <= hd; i += 64) {  // 8 x 16-byte loads in flight; products still added in k order
    i32x4 kv[8];
#pragma unroll
    for (int u = 0; u < 8; u++) kv[u] = *(const i32x4*)(kr + i + 8 * u);
#pragma unroll
    for (int u = 0; u < 8; u++)
#pragma unroll
      for (int w4 = 0; w4 < 4; w4++) {
        const unsigned w = (unsigned)kv[u][w4];
        acc += qg[i + 8 * u + 2 * w4] * h2f((unsigned short)(w & 0xffffu));
        acc += qg[i + 8 * u + 2 * w4 + 1] * h2f((unsigned short)(w >> 16));
      }
  }
  for (; i + 8 <= hd; i += 8) {
    const i32x4 kv = *(const i32x4*)(kr + i);
#pragma unroll
    for (int w4 = 0; w4 < 4; w4++) {
      const unsigned w = (unsigned)kv[w4];
      acc += qg[i + 2 * w4] * h2f((unsigned short)(w & 0xffffu));
      acc += qg[i + 2 * w4 + 1] * h2f((unsigned short)(w >> 16));
    }
  }
  for (; i < hd; i++) acc += qg[i] * h2f(kr[i]);
  scores_g[(size_t)(j * G + g) * seq_cap + t] = acc;
}

__global__ __launch_bounds__(256) void k_attn_softmax(const float* __restrict__ scores_g, const int* __restrict__ pos_d,
                                                      const unsigned short* __restrict__ exp_tab,
                                                      unsigned short* __restrict__ p16, int seq_cap) {
  extern __shared__ float lds[];
  __shared__ float s_red[4];
  __shared__ float s_val;
  const int head = blockIdx.x, seq = *pos_d + 1;
  for (int t = threadIdx.x; t < seq; t += blockDim.x) lds[t] = scores_g[(size_t)head * seq_cap + t];
  __syncthreads();
  softmax_row<true>(lds, seq, exp_tab, s_red, &s_val);
  for (int t = threadIdx.x; t < seq; t += blockDim.x) p16[(size_t)head * seq_cap + t] = f2h(lds[t]);  // exact: already f16 values
}

typedef _Float16 h16x2 __attribute__((ext_vector_type(2)));
#define ATTN_PV_TILE 256
#define ATTN_PV_ROW (ATTN_PV_TILE + 4)  // words per LDS row: 16-byte aligned rows, shifted by 4 banks from each other
template <int G>
__global__ __launch_bounds__(256) void k_attn_pv(const unsigned short* __restrict__ p16, const unsigned short* __restrict__ vc,
                                                 const int* __restrict__ pos_d, float* __restrict__ out,
                                                 signed char* __restrict__ xq, unsigned short* __restrict__ xd,
                                                 void* __restrict__ xisum, int hd, int seq_cap, int q81) {
  constexpr int T = ATTN_PV_TILE, ROW = ATTN_PV_ROW;
  // LDS, two buffers each: V tile transposed to [16 dim pairs][T] words (a chain lane reads 4 consecutive positions
  // of its dim pair with one ds_read_b128), P tile [G][T] words holding {p, p} (the packed multiplier, ready-made)
  __shared__ __attribute__((aligned(16))) unsigned vt[2][16 * ROW];
  __shared__ __attribute__((aligned(16))) unsigned pt[2][G * ROW];
  const int tid = threadIdx.x;
  const int nslice = hd / 32;
  const int j = blockIdx.x / nslice, sl = blockIdx.x % nslice;
  const int seq = *pos_d + 1;
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

// ---- GEMV + residual: x[row] = W[row].xq + x[row]   (matmul_vec, then add_inplace: arithmetic.rs:27-33) ---
template <int FMT, int R, bool ADD>  // ADD: x[row] += W.xq (residual); else out[row] = W.xq (tensor-parallel partial sum)
__global__ __launch_bounds__(128) void k_gemv_res(Planes w, typename ActOf<FMT>::type act, float* __restrict__ x, int m, int nb) {
  const int lane = threadIdx.x & 63;
  const int wave = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
  const int row0 = wave * R;
  if (row0 >= m) return;
  // the residual is loaded up front (its latency overlaps the weight stream instead of trailing the reduction)
  float res[R];
#pragma unroll
  for (int r = 0; r < R; r++) res[r] = (ADD && lane == 0 && row0 + r < m) ? x[row0 + r] : 0.f;
  float acc[R];
  rows_dot<FMT, R>(w.q, w.d, act, row0, m, nb, lane, acc);
#pragma unroll
  for (int r = 0; r < R; r++) {
    float s = wave_sum_f32(acc[r]);
    if (lane == 0 && row0 + r < m) x[row0 + r] = ADD ? s + res[r] : s;
  }
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
    qs[e] = KV16 ? h2f(f2h(v)) : v;
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
          for (int i = 0; i < hd; i += 16) {  // hd % 16 == 0 (host check); products added in k order per query
            const i32x4 k0 = *(const i32x4*)(kr + i), k1 = *(const i32x4*)(kr + i + 8);
            float kf[16];
#pragma unroll
            for (int u = 0; u < 4; u++) {
              kf[2 * u] = h2f((unsigned short)((unsigned)k0[u] & 0xffffu));
              kf[2 * u + 1] = h2f((unsigned short)((unsigned)k0[u] >> 16));
              kf[8 + 2 * u] = h2f((unsigned short)((unsigned)k1[u] & 0xffffu));
              kf[8 + 2 * u + 1] = h2f((unsigned short)((unsigned)k1[u] >> 16));
            }
#pragma unroll
            for (int u = 0; u < QW; u++) {
              const f32x4* qp = (const f32x4*)(qs + (q0 + u) * hd + i);
#pragma unroll
              for (int v4 = 0; v4 < 4; v4++) {
                const f32x4 qv = qp[v4];
                acc[u] += qv[0] * kf[4 * v4];
                acc[u] += qv[1] * kf[4 * v4 + 1];
                acc[u] += qv[2] * kf[4 * v4 + 2];
                acc[u] += qv[3] * kf[4 * v4 + 3];
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

__global__ __launch_bounds__(256) void k_res_epi(const float* __restrict__ tmp, float* __restrict__ x, int m, int add) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < m) x[i] = add ? tmp[i] + x[i] : tmp[i];
}
// single-device simulation of the tensor-parallel all-reduce: every rank's partial <- sum over ranks (rank order)
struct SimPtrs {
  float* p[8];
};
__global__ __launch_bounds__(256) void k_sim_allreduce(SimPtrs ptrs, int nranks, int n) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  float s = ptrs.p[0][i];
  for (int r = 1; r < nranks; r++) s += ptrs.p[r][i];
  for (int r = 0; r < nranks; r++) ptrs.p[r][i] = s;
}

// ---- fast mode: GEMV + residual with the NEXT RMSNorm + quantization done in the epilogue ------------------------
// The separate norm+quantize launch is a single-workgroup latency stage (6 us x 65 per token on Llama-3-8B).  Here
// the producer of x (wo / ffn_down + residual) finishes the job: a 1024-thread workgroup owns 32 consecutive rows
// = one rmsnorm chunk = one Q8_0 block (the k_gateup_q shape).  It publishes its ordered chunk sum of squares as
// one 8-byte {sum, epoch} granule (a single write-through store: data and tag travel together, no fence needed),
// gathers all dim/32 granules (one wave polls them with relaxed agent-scope loads), adds them in chunk order like
// rms_norm.rs:35-40, and normalizes + quantizes its own block.  Every bit of the result equals k_norm_quant's:
// same chunk sums, same serial chain, same divisions.  All dim/32 workgroups are co-resident by construction
// (<= one per CU, checked at create); the poll is bounded and raises `fault` instead of hanging.
// Q8_K quantizer of an f32 vector straight into LDS planes (q | d | bsums, as stage_act_q8k lays them out): one
// wave per super-block.  The Q4_K wo / ffn_down kernels run it as their prologue on the attention output / h,
// each workgroup for itself (16 KB / 56 KB of L2 reads), instead of a quantizer launch in front of them.
__device__ __forceinline__ void stage_quant_q8k(const float* __restrict__ x, int nsb, unsigned* sq, float* sd, short* sbs) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
  // four super-blocks of loads in flight per wave (ffn_down: 56 super-blocks over 16 waves; a round is one L2 latency)
  for (int sb0 = wave; sb0 < nsb; sb0 += 4 * nw) {
    f32x4 v[4];
#pragma unroll
    for (int u = 0; u < 4; u++) {
      const int sb = sb0 + u * nw;
      v[u] = ((const f32x4*)x)[(sb < nsb ? sb : sb0) * 64 + lane];
    }
#pragma unroll
    for (int u = 0; u < 4; u++) {
      const int sb = sb0 + u * nw;
      if (sb >= nsb) break;  // wave-uniform
      const Q8KLane o = q8k_wave_quant(v[u], lane);
      sq[sb * 64 + lane] = o.packed;
      if ((lane & 3) == 0) sbs[sb * 16 + (lane >> 2)] = (short)o.quad_sum;
      if (lane == 0) sd[sb] = o.d;
    }
  }
  __syncthreads();
}

struct NormGather {
  unsigned long long* slots;  // dim/16 granules: each workgroup's ordered sum of squares over its rows
  unsigned long long* pair;   // dim row granules (read by a split chunk's partner / a Q8_K super-block's neighbours)
  const int* serial;          // decode-step serial number (never reset): makes the epoch unique per launch
  int* fault;
  int nseg, seg;
};
__device__ __forceinline__ unsigned long long ld_granule(const unsigned long long* p) {
  return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
// SPLIT workgroups share one 32-row chunk (16 waves x 2 / SPLIT rows); QIN (Q4_K): the rhs is the f32 vector xin
// The tail of the wo / ffn_down kernels (k_gemv_res_nq, k_ffn): acc[] = this wave's RW row dots.  Publishes the
// workgroup's rows / sum of squares, takes the one in-launch hop, normalizes + quantizes the rows it owns.
// wg_index / nwg_all: this workgroup's index among the SPLIT * nchunks workgroups of the stage.
template <int FMT, int SPLIT>
__device__ __forceinline__ void nq_epilogue(float (&acc)[2 / SPLIT], float res, float wn, f32x4 wn4, unsigned epoch, float* hv,
                                            float* __restrict__ x, signed char* __restrict__ q, void* __restrict__ d,
                                            void* __restrict__ isum, const NormGather& ng, float eps, int blk, int part, int nchunks,
                                            int row, int lane, int wave, int wg_index, int nwg_all) {
  constexpr bool Q81 = FMT == CRABML_HIP_Q4_1;
  constexpr bool KQ = FMT == CRABML_HIP_Q4_K;
  constexpr int RW = 2 / SPLIT;
  constexpr int ROWS = 32 / SPLIT;
  // ---- epilogue: publish, one in-launch hop, normalize + quantize -------------------------------------------
  // every row goes out as a {value, epoch} granule when another workgroup needs it (the partner of a split chunk;
  // the seven neighbours of a Q8_K super-block), the workgroup's ordered sum of squares as one more
  constexpr bool ROWG = SPLIT > 1 || KQ;
#pragma unroll
  for (int r = 0; r < RW; r++) {
    const float s = wave_sum_f32(acc[r]);
    if (lane == 0) hv[part * ROWS + wave * RW + r] = s;
  }
  __syncthreads();
  if (wave != 0) return;
  // wave 0 owns the stores: ROWS consecutive rows per instruction (x and the row granules are one or two lines,
  // not 32 separate partial writes from 16 waves)
  if (lane < ROWS) {
    const float xv = hv[part * ROWS + lane] + res;  // x = matmul_out + x (llama2.rs:266 / :636)
    x[row + lane] = xv;
    hv[part * ROWS + lane] = xv;
    if (ROWG)
      __hip_atomic_store(ng.pair + row + lane, ((unsigned long long)epoch << 32) | (unsigned long long)__builtin_bit_cast(unsigned, xv),
                         __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
  __builtin_amdgcn_wave_barrier();
  // sum of squares of a chunk = (rows 0..15 in order) + (rows 16..31 in order): a split chunk's two workgroups
  // each own one half (norm_quant_block<HALF> computes the same)
  float cs;
  {
    float h0 = -0.0f, h1 = -0.0f;
#pragma unroll
    for (int j = 0; j < 4; j++) {
      const f32x4 t = ((const f32x4*)hv)[(SPLIT > 1 ? part * 4 : 0) + j];
      h0 += t[0] * t[0];
      h0 += t[1] * t[1];
      h0 += t[2] * t[2];
      h0 += t[3] * t[3];
    }
    if (SPLIT == 1) {
#pragma unroll
      for (int j = 4; j < 8; j++) {
        const f32x4 t = ((const f32x4*)hv)[j];
        h1 += t[0] * t[0];
        h1 += t[1] * t[1];
        h1 += t[2] * t[2];
        h1 += t[3] * t[3];
      }
      cs = h0 + h1;
    } else {
      cs = h0;
    }
  }
  if (lane == 0)
    __hip_atomic_store(ng.slots + wg_index, ((unsigned long long)epoch << 32) | (unsigned long long)__builtin_bit_cast(unsigned, cs),
                       __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  __builtin_amdgcn_s_waitcnt(0x0F70);  // vmcnt(0): the granule is on its way before the polls queue up behind it
  auto poll = [&](const unsigned long long* p) -> float {
    unsigned long long g = ld_granule(p);
    int tries = 0;
    while ((unsigned)(g >> 32) != epoch && tries < (1 << 21)) {
      __builtin_amdgcn_s_sleep(2);
      g = ld_granule(p);
      tries++;
    }
    if ((unsigned)(g >> 32) != epoch) *ng.fault = 1;  // a workgroup never arrived: flagged, not hung
    return __builtin_bit_cast(float, (unsigned)g);
  };
  // the rows of other workgroups first (published before their sums; the loads fly while the stragglers arrive) ...
  const int l32 = lane & 31;
  const bool own = l32 >= part * ROWS && l32 < (part + 1) * ROWS;
  const int sb = blk >> 3;
  float v = 0.0f;
  f32x4 v4 = {0.f, 0.f, 0.f, 0.f};
  if constexpr (KQ) {
    const unsigned long long* p = ng.pair + sb * 256 + lane * 4;
    unsigned long long g[4];
#pragma unroll
    for (int i = 0; i < 4; i++) g[i] = ld_granule(p + i);
#pragma unroll
    for (int i = 0; i < 4; i++) v4[i] = (unsigned)(g[i] >> 32) == epoch ? __builtin_bit_cast(float, (unsigned)g[i]) : poll(p + i);
  } else if (SPLIT > 1) {
    if (lane < 32) v = own ? hv[l32] : poll(ng.pair + blk * 32 + l32);
  } else {
    v = hv[l32];
  }
  // ... then the hop: every workgroup's sum, added strictly in chunk order
  float sum = 0.0f;
  const int nwg = nwg_all;
  for (int base = 0; base < nwg; base += 64) {
    const int c = base + lane;
    float cv = c < nwg ? poll(ng.slots + c) : 0.0f;
    if (SPLIT > 1) cv += dpp_f<0xB1>(cv);  // chunk = its two halves (the same value on both lanes of the pair)
#pragma unroll
    for (int i = 0; i < 64; i += SPLIT) sum += rl_f(cv, i);  // lanes past the grid add +0.0
  }
  const float rms = sqrtf(sum / (float)(nchunks * 32) + eps);
  if constexpr (!KQ) {
    const float xn = (v / rms) * wn;
    const QLane o = quant_lane32<Q81>(xn, true);
    if (lane < 32 && own) {
      q[blk * 32 + lane] = o.q;
      if (lane == 0) {
        ((unsigned short*)d)[blk] = o.d;
        store_qaux<Q81>(isum, blk, o.aux);
      }
    }
  } else {
    // Q8_K (buf_q8_k.rs:84-131): the scale comes from the FIRST element of maximal |x| of the 256-element
    // super-block = this chunk and its 7 neighbours.  The wave holds the super-block's 256 rows (4 per lane, from
    // their granules), normalizes them all and runs the whole block's quantizer; it stores the part that is its own.
    f32x4 xn;
#pragma unroll
    for (int i = 0; i < 4; i++) xn[i] = (v4[i] / rms) * wn4[i];
    const Q8KLane o = q8k_wave_quant(xn, lane);
    const int l0 = (blk & 7) * 8 + part * (ROWS / 4);
    if (lane >= l0 && lane < l0 + ROWS / 4) {
      ((unsigned*)q)[sb * 64 + lane] = o.packed;
      if ((lane & 3) == 0) ((short*)isum)[sb * 16 + (lane >> 2)] = (short)o.quad_sum;
    }
    if (lane == 0 && (blk & 7) == 0 && part == 0) ((float*)d)[sb] = o.d;
  }
}

template <int FMT, int SPLIT, bool QIN = false>
__global__ __launch_bounds__(1024) void k_gemv_res_nq(Planes w, typename ActOf<FMT>::type act, const float* __restrict__ xin,
                                                      float* __restrict__ x,
                                                      const float* __restrict__ wnext, float eps,
                                                      signed char* __restrict__ q, void* __restrict__ d,
                                                      void* __restrict__ isum, NormGather ng, int nb, Planes6 w6) {
  constexpr bool KQ = FMT == CRABML_HIP_Q4_K;  // Q4_K weights: nb counts super-blocks, the output is Q8_K
  constexpr int RW = 2 / SPLIT;         // rows per wave
  constexpr int ROWS = 32 / SPLIT;      // rows per workgroup
  __shared__ __attribute__((aligned(16))) float hv[32];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int blk = blockIdx.x / SPLIT, part = blockIdx.x % SPLIT;
  const int nchunks = gridDim.x / SPLIT;
  const int row = blk * 32 + part * ROWS + wave * RW;
  float res = 0.f;                         // wave 0: the residual of row (first row of the workgroup) + lane
  float wn = 0.f;                          // the next RMSNorm's weights for the rows this wave will normalize,
  f32x4 wn4 = {0.f, 0.f, 0.f, 0.f};        // loaded up front (off the critical path after the hop)
  unsigned epoch = 0;
  if (wave == 0) {
    if (lane < ROWS) res = x[row + lane];
    if constexpr (KQ)
      wn4 = ((const f32x4*)wnext)[(blk >> 3) * 64 + lane];
    else
      wn = wnext[blk * 32 + (lane & 31)];
  }
  if (wave == 0) epoch = (unsigned)(*ng.serial) * (unsigned)ng.nseg + (unsigned)ng.seg + 1u;
  // RW rows x two blocks per lane in flight (one workgroup per CU: the loads have to supply the parallelism);
  // terms are added in block order, as rows_partial does
  float acc[RW];
  if constexpr (KQ && QIN) {
    // the rhs arrives as f32 (attention output / h): quantize it to Q8_K in LDS first
    extern __shared__ i32x4 lds_act[];  // q[k] | d[k/256] f32 | bsums[k/16] i16
    float* sd = (float*)(lds_act + nb * 16);
    short* sbs = (short*)(sd + nb);
    // the first weight pieces are requested before the prologue (they do not depend on it): its L2 round trip
    // and the quantizer run under the HBM latency of the stream's head
    if (w6.base != nullptr) {  // this layer's matrix is Q6_K (a *_K_M mix): same rhs, its own inner loop
      stage_quant_q8k(xin, nb, (unsigned*)lds_act, sd, sbs);
      const ActQ8_K la6{lds_act, sd, sbs};
      rows_partial_q6k<RW>(w6.base, w6.off_qh, la6, row, nchunks * 32, nb, lane, acc);
      nq_epilogue<FMT, SPLIT>(acc, res, wn, wn4, epoch, hv, x, q, d, isum, ng, eps, blk, part, nchunks, row, lane, wave,
                              (int)blockIdx.x, (int)gridDim.x);
      return;
    }
    constexpr int PRE = 2;
    Q4KPiece<false> pw[PRE][RW];
#pragma unroll
    for (int it = 0; it < PRE; it++) {
      const int c = it * 64 + lane;
#pragma unroll
      for (int r = 0; r < RW; r++) pw[it][r] = q4k_load<false>(w.q, (const i32x4*)w.d, (size_t)(row + r), nb, c < nb * 8 ? c : nb * 8 - 1, lane);
    }
    stage_quant_q8k(xin, nb, (unsigned*)lds_act, sd, sbs);
    const ActQ8_K la{lds_act, sd, sbs};
#pragma unroll
    for (int r = 0; r < RW; r++) acc[r] = 0.f;
#pragma unroll
    for (int it = 0; it < PRE; it++) {
      const int c = it * 64 + lane;
      if (c < nb * 8) {
        const Q4KX xx = q4k_loadx(la, c);
#pragma unroll
        for (int r = 0; r < RW; r++) acc[r] += q4k_term<false>(pw[it][r], xx, c);
      }
    }
    rows_partial_q4k<RW, false>(w.q, (const i32x4*)w.d, la, row, nchunks * 32, nb, lane, acc, PRE * 64);
  } else if constexpr (KQ) {
    if (w6.base != nullptr)
      rows_partial_q6k<RW>(w6.base, w6.off_qh, act, row, nchunks * 32, nb, lane, acc);
    else
      rows_partial_q4k<RW>(w.q, (const i32x4*)w.d, act, row, nchunks * 32, nb, lane, acc);
  } else {
    using F = BlockFmt<FMT>;
#pragma unroll
    for (int r = 0; r < RW; r++) acc[r] = 0.f;
    const int nu = nb * F::UNITS;
    for (int u = lane; u < nu; u += 128) {
      const int u2 = u + 64;
      const bool two = u2 < nu;
      const int uu = two ? u2 : u;
      typename F::Blk ka[RW], kb[RW];
#pragma unroll
      for (int r = 0; r < RW; r++) {
        ka[r] = F::load(w.q, w.d, (size_t)(row + r), nb, u);
        kb[r] = F::load(w.q, w.d, (size_t)(row + r), nb, uu);
      }
      const XUnit xa = F::loadx(act, u), xb = F::loadx(act, uu);
#pragma unroll
      for (int r = 0; r < RW; r++) acc[r] += F::term(ka[r], xa);
      if (two) {
#pragma unroll
        for (int r = 0; r < RW; r++) acc[r] += F::term(kb[r], xb);
      }
    }
  }
  nq_epilogue<FMT, SPLIT>(acc, res, wn, wn4, epoch, hv, x, q, d, isum, ng, eps, blk, part, nchunks, row, lane, wave, (int)blockIdx.x,
                          (int)gridDim.x);
}

// ---- gate/up GEMV + SiLU * mul: h[i] = silu(Wg[i].xq) * (Wu[i].xq)   (silu.rs:6-13, arithmetic.rs:57-66) ---
__device__ __forceinline__ float silu_mul(float g, float u, const unsigned short* __restrict__ exp_tab) {
  float nexp = exp_cached_f(-g, exp_tab);
  return (g / (1.0f + nexp)) * u;
}
template <int FMT>
__global__ __launch_bounds__(128) void k_gateup(Planes wg, Planes wu, typename ActOf<FMT>::type act,
                                                const unsigned short* __restrict__ exp_tab, float* __restrict__ h, int m, int nb) {
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
  if (row >= m) return;
  float ag[1], au[1];
  rows_dot<FMT, 1>(wg.q, wg.d, act, row, m, nb, lane, ag);
  rows_dot<FMT, 1>(wu.q, wu.d, act, row, m, nb, lane, au);
  const float g = wave_sum_f32(ag[0]), u = wave_sum_f32(au[0]);
  if (lane == 0) h[row] = silu_mul(g, u, exp_tab);
}
// Q4_K gate/up with the Q8_K activation planes staged in LDS once per workgroup (1024 threads = 32 hidden rows x
// {gate, up}): the per-lane activation reads (2 x 16 B + d + 2 bsums per 16 B of quants) leave the vector-memory
// path, which the K-quant inner loop otherwise keeps ~57 % busy (rocprofv3 TA_BUSY) while VALU sits at 15 %.
__global__ __launch_bounds__(1024) void k_gateup_k_lds(Planes wg, Planes wu, ActQ8_K act, const unsigned short* __restrict__ exp_tab,
                                                       float* __restrict__ h, int m, int nsb) {
  extern __shared__ i32x4 lds_act[];  // q[k] | d[k/256] f32 | bsums[k/16] i16
  const int k = nsb * 256;
  i32x4* sq = lds_act;
  float* sd = (float*)(sq + k / 16);
  short* sbs = (short*)(sd + nsb);
  for (int i = threadIdx.x; i < k / 16; i += 1024) sq[i] = act.q[i];
  for (int i = threadIdx.x; i < nsb; i += 1024) sd[i] = act.d[i];
  for (int i = threadIdx.x; i < k / 16; i += 1024) sbs[i] = act.bsums[i];
  __syncthreads();
  const ActQ8_K la{sq, sd, sbs};
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int row0 = blockIdx.x * 32 + wave * 2;
  if (row0 >= m) return;
  float ag[2], au[2];
  rows_partial_q4k<2, false>(wg.q, (const i32x4*)wg.d, la, row0, m, nsb, lane, ag);
  rows_partial_q4k<2, false>(wu.q, (const i32x4*)wu.d, la, row0, m, nsb, lane, au);
#pragma unroll
  for (int r = 0; r < 2; r++) {
    const float g = wave_sum_f32(ag[r]), u = wave_sum_f32(au[r]);
    if (lane == 0 && row0 + r < m) h[row0 + r] = silu_mul(g, u, exp_tab);
  }
}

// Same, with the Q8_0 quantization of h (the rhs of ffn_down) folded in: a 1024-thread workgroup owns 32
// consecutive hidden rows = one quant block; each of its 16 waves computes 2 rows (4 weight rows in flight),
// parks the h values in LDS, and one half-wave quantizes the block (buf_q8_0.rs:87-134).  hidden/32
// workgroups (448 for Llama-3-8B) are all resident at once (2 per CU).  Saves a launch per layer.
template <int FMT>
__global__ __launch_bounds__(1024) void k_gateup_q(Planes wg, Planes wu, typename ActOf<FMT>::type act,
                                                   const unsigned short* __restrict__ exp_tab, signed char* __restrict__ q,
                                                   unsigned short* __restrict__ d, void* __restrict__ isum, int nb) {
  using F = BlockFmt<FMT>;
  constexpr bool Q81 = FMT == CRABML_HIP_Q4_1;
  __shared__ float hv[32];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int blk = blockIdx.x;
  const int row = blk * 32 + wave * 2;  // rows row, row+1
  float g0 = 0.f, g1 = 0.f, u0 = 0.f, u1 = 0.f;
  const int nu = nb * F::UNITS;
  for (int u = lane; u < nu; u += 64) {
    typename F::Blk bg0 = F::load(wg.q, wg.d, (size_t)row, nb, u);
    typename F::Blk bu0 = F::load(wu.q, wu.d, (size_t)row, nb, u);
    typename F::Blk bg1 = F::load(wg.q, wg.d, (size_t)row + 1, nb, u);
    typename F::Blk bu1 = F::load(wu.q, wu.d, (size_t)row + 1, nb, u);
    const XUnit x = F::loadx(act, u);
    g0 += F::term(bg0, x);
    u0 += F::term(bu0, x);
    g1 += F::term(bg1, x);
    u1 += F::term(bu1, x);
  }
  g0 = wave_sum_f32(g0);
  u0 = wave_sum_f32(u0);
  g1 = wave_sum_f32(g1);
  u1 = wave_sum_f32(u1);
  if (lane == 0) {
    hv[wave * 2] = silu_mul(g0, u0, exp_tab);
    hv[wave * 2 + 1] = silu_mul(g1, u1, exp_tab);
  }
  __syncthreads();
  if (threadIdx.x < 32) {
    const QLane o = quant_lane32<Q81>(hv[threadIdx.x], true);
    q[blk * 32 + threadIdx.x] = o.q;
    if (threadIdx.x == 0) {
      d[blk] = o.d;
      store_qaux<Q81>(isum, blk, o.aux);
    }
  }
}
// ---- gate/up + SiLU*mul + quantize + ffn_down + residual + next RMSNorm/quantize in ONE launch ----------------------
// EXPERIMENT, opt-in (CRABML_HIP_LLAMA_FFN_FUSION): measured 27.5-28.7 us against 12.9 + 0.8 + 10.7 us for the two
// kernels it replaces on the 8B shape ("measured and rejected"), bit-identical to them.
// The two halves of the FFN are k_gateup_q and k_gemv_res_nq<FMT, 2> back to back; what the single launch was meant
// to buy is the boundary between them: ffn_down's first weight loads are requested BEFORE its workgroup waits for h,
// so the HBM round trip of the stream's head runs under the hand-off instead of after a kernel boundary.  h never touches a
// plane in global memory: every 32-row block goes out as 8 {4 quants, epoch} granules + 1 {d | aux, epoch} granule
// (aux = the block's quant sum for Q8_0 -- |sum| <= 4096 fits 16 bits -- or s for Q8_1), and every workgroup polls
// all of them (hidden/4 + hidden/32 relaxed agent-scope loads, 4 per thread) into its own LDS copy of the planes.
// Grid = dim/16 workgroups of 1024 threads, all resident (the norm-epilogue condition); workgroup b owns the
// hidden blocks b and b + grid (the latter when it exists) and, for ffn_down, half of chunk b / 2.
struct HGather {
  unsigned long long* hq;  // hidden/4 granules
  unsigned long long* hs;  // hidden/32 granules
};
template <class F, int NB, class ACT>
__device__ __forceinline__ void ffn_gateup_rows(const Planes& wg, const Planes& wu, const ACT& act, int nb, int lane,
                                                const int (&row)[NB], float (&g)[NB][2], float (&u2)[NB][2]) {
#pragma unroll
  for (int k = 0; k < NB; k++) g[k][0] = g[k][1] = u2[k][0] = u2[k][1] = 0.f;
  const int nu = nb * F::UNITS;
  // two units per row in flight (one workgroup per CU: the loads have to supply the parallelism; with one unit per
  // iteration a wave paid an HBM round trip per iteration and the phase streamed at 3.3 TB/s); terms in block order
  for (int u = lane; u < nu; u += 128) {
    const int ub = u + 64;
    const bool two = ub < nu;
    const int uu = two ? ub : u;
    typename F::Blk bg[NB][2][2], bu[NB][2][2];
#pragma unroll
    for (int k = 0; k < NB; k++)
#pragma unroll
      for (int r = 0; r < 2; r++) {
        bg[k][r][0] = F::load(wg.q, wg.d, (size_t)(row[k] + r), nb, u);
        bu[k][r][0] = F::load(wu.q, wu.d, (size_t)(row[k] + r), nb, u);
        bg[k][r][1] = F::load(wg.q, wg.d, (size_t)(row[k] + r), nb, uu);
        bu[k][r][1] = F::load(wu.q, wu.d, (size_t)(row[k] + r), nb, uu);
      }
    const XUnit xa = F::loadx(act, u), xb = F::loadx(act, uu);
#pragma unroll
    for (int k = 0; k < NB; k++)
#pragma unroll
      for (int r = 0; r < 2; r++) {
        g[k][r] += F::term(bg[k][r][0], xa);
        u2[k][r] += F::term(bu[k][r][0], xa);
      }
    if (two) {
#pragma unroll
      for (int k = 0; k < NB; k++)
#pragma unroll
        for (int r = 0; r < 2; r++) {
          g[k][r] += F::term(bg[k][r][1], xb);
          u2[k][r] += F::term(bu[k][r][1], xb);
        }
    }
  }
#pragma unroll
  for (int k = 0; k < NB; k++)
#pragma unroll
    for (int r = 0; r < 2; r++) {
      g[k][r] = wave_sum_f32(g[k][r]);
      u2[k][r] = wave_sum_f32(u2[k][r]);
    }
}
template <int FMT>
__global__ __launch_bounds__(1024) void k_ffn(Planes wg, Planes wu, Planes wdn, typename ActOf<FMT>::type act,
                                              const unsigned short* __restrict__ exp_tab, float* __restrict__ x,
                                              const float* __restrict__ wnext, float eps, signed char* __restrict__ q,
                                              void* __restrict__ d, void* __restrict__ isum, NormGather ng, HGather hg, int nb_in,
                                              int nblk_h, int off_d, int off_aux) {
  using F = BlockFmt<FMT>;
  constexpr bool Q81 = FMT == CRABML_HIP_Q4_1;
  extern __shared__ i32x4 lds_h[];  // phase B: h's activation planes, act_layout order
  __shared__ __attribute__((aligned(16))) float hv[64];
  __shared__ __attribute__((aligned(16))) signed char hqb[64];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int G = (int)gridDim.x;
  const unsigned epoch = (unsigned)(*ng.serial) * (unsigned)ng.nseg + (unsigned)ng.seg + 1u;
  // ---- phase A: gate/up rows of this workgroup's hidden blocks (wave w: rows 2w, 2w + 1 of each block)
  const int b0 = (int)blockIdx.x, b1 = b0 + G;
  const bool has0 = b0 < nblk_h, has1 = b1 < nblk_h;
  if (has0) {
    if (has1) {
      const int row[2] = {b0 * 32 + wave * 2, b1 * 32 + wave * 2};
      float g[2][2], u2[2][2];
      ffn_gateup_rows<F, 2>(wg, wu, act, nb_in, lane, row, g, u2);
      if (lane == 0) {
#pragma unroll
        for (int k = 0; k < 2; k++)
#pragma unroll
          for (int r = 0; r < 2; r++) hv[k * 32 + wave * 2 + r] = silu_mul(g[k][r], u2[k][r], exp_tab);
      }
    } else {
      const int row[1] = {b0 * 32 + wave * 2};
      float g[1][2], u2[1][2];
      ffn_gateup_rows<F, 1>(wg, wu, act, nb_in, lane, row, g, u2);
      if (lane == 0) {
        hv[wave * 2] = silu_mul(g[0][0], u2[0][0], exp_tab);
        hv[wave * 2 + 1] = silu_mul(g[0][1], u2[0][1], exp_tab);
      }
    }
  }
  __syncthreads();
  if ((wave == 0 && has0) || (wave == 1 && has1)) {  // wave k quantizes and publishes block k (buf_q8_0.rs:87-134)
    const int hb = wave == 0 ? b0 : b1;
    const QLane o = quant_lane32<Q81>(hv[wave * 32 + (lane & 31)], true);
    if (lane < 32) hqb[wave * 32 + lane] = o.q;
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __builtin_amdgcn_wave_barrier();
    if (lane < 8)
      __hip_atomic_store(hg.hq + hb * 8 + lane, ((unsigned long long)epoch << 32) | (unsigned long long)((const unsigned*)hqb)[wave * 8 + lane],
                         __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (lane == 0)
      __hip_atomic_store(hg.hs + hb, ((unsigned long long)epoch << 32) | (unsigned long long)((unsigned)o.d | (((unsigned)o.aux & 0xffffu) << 16)),
                         __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __builtin_amdgcn_s_waitcnt(0x0F70);  // vmcnt(0): on their way before this wave starts polling
  }
  // ---- phase B set-up: ffn_down row of this wave, its first weight units requested before the hand-off
  const int blk = (int)blockIdx.x >> 1, part = (int)blockIdx.x & 1;
  const int nchunks = G >> 1;
  const int row = blk * 32 + part * 16 + wave;
  float res = 0.f, wn = 0.f;
  const f32x4 wn4 = {0.f, 0.f, 0.f, 0.f};
  if (wave == 0) {
    if (lane < 16) res = x[row + lane];
    wn = wnext[blk * 32 + (lane & 31)];
  }
  const int nu = nblk_h * F::UNITS;
  const int ua = lane < nu ? lane : nu - 1, ub = lane + 64 < nu ? lane + 64 : nu - 1;
  const typename F::Blk ka0 = F::load(wdn.q, wdn.d, (size_t)row, nblk_h, ua);
  const typename F::Blk kb0 = F::load(wdn.q, wdn.d, (size_t)row, nblk_h, ub);
  // ---- the hand-off: all of h into this workgroup's LDS planes
  char* P = (char*)lds_h;
  auto poll = [&](const unsigned long long* p) -> unsigned {
    unsigned long long gq = ld_granule(p);
    int tries = 0;
    while ((unsigned)(gq >> 32) != epoch && tries < (1 << 21)) {
      __builtin_amdgcn_s_sleep(2);
      gq = ld_granule(p);
      tries++;
    }
    if ((unsigned)(gq >> 32) != epoch) *ng.fault = 1;  // a workgroup never arrived: flagged, not hung
    return (unsigned)gq;
  };
  // every thread requests its (up to 4) quant granules right away -- for the workgroup that arrives last, which
  // sets the pace, everything is already published and comes back fresh in the same round trip as the scale
  // granules; wave 0 alone spins on the scale granules (1024 spinning threads per early workgroup would sit on the
  // memory path the late workgroups are still streaming weights through); stale quant granules are re-polled after
  const int nq = nblk_h * 8;
  unsigned long long gq[4];
#pragma unroll
  for (int u = 0; u < 4; u++) {
    const int i = tid + u * 1024;
    gq[u] = ld_granule(hg.hq + (i < nq ? i : tid));
  }
  if (wave == 0) {
    for (int i = lane; i < nblk_h; i += 64) {
      const unsigned v = poll(hg.hs + i);
      ((unsigned short*)(P + off_d))[i] = (unsigned short)(v & 0xffffu);
      if constexpr (Q81)
        ((unsigned short*)(P + off_aux))[i] = (unsigned short)(v >> 16);
      else
        ((int*)(P + off_aux))[i] = (int)(short)(v >> 16);
    }
  }
  __syncthreads();
#pragma unroll
  for (int u = 0; u < 4; u++) {
    const int i = tid + u * 1024;
    if (i < nq) ((unsigned*)P)[i] = (unsigned)(gq[u] >> 32) == epoch ? (unsigned)gq[u] : poll(hg.hq + i);
  }
  for (int i = tid + 4 * 1024; i < nq; i += 1024) ((unsigned*)P)[i] = poll(hg.hq + i);  // hidden > 16384 only
  __syncthreads();
  // ---- phase B: the ffn_down row against the LDS planes (terms in block order, as k_gemv_res_nq adds them)
  typename ActOf<FMT>::type la;
  la.q = (const i32x4*)P;
  la.d = (const unsigned short*)(P + off_d);
  if constexpr (Q81)
    la.s = (const unsigned short*)(P + off_aux);
  else
    la.isum = (const int*)(P + off_aux);
  float acc[1] = {0.f};
  {
    const XUnit xa = F::loadx(la, ua), xb = F::loadx(la, ub);
    if (lane < nu) acc[0] += F::term(ka0, xa);
    if (lane + 64 < nu) acc[0] += F::term(kb0, xb);
  }
  for (int u = lane + 128; u < nu; u += 128) {
    const int u2 = u + 64;
    const bool two = u2 < nu;
    const int uu = two ? u2 : u;
    const typename F::Blk ka = F::load(wdn.q, wdn.d, (size_t)row, nblk_h, u);
    const typename F::Blk kb = F::load(wdn.q, wdn.d, (size_t)row, nblk_h, uu);
    const XUnit xa = F::loadx(la, u), xb = F::loadx(la, uu);
    acc[0] += F::term(ka, xa);
    if (two) acc[0] += F::term(kb, xb);
  }
  nq_epilogue<FMT, 2>(acc, res, wn, wn4, epoch, hv, x, q, d, isum, ng, eps, blk, part, nchunks, row - wave, lane, wave,
                      (int)blockIdx.x, G);
}

__global__ __launch_bounds__(256) void k_gateup_epi(const float* __restrict__ g, const float* __restrict__ u,
                                                    const unsigned short* __restrict__ exp_tab, float* __restrict__ h, int m) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < m) h[i] = silu_mul(g[i], u[i], exp_tab);
}

// ---- greedy sampler + advance: Iterator::max_by keeps the LAST maximum (sampler.rs:109-116) ------------
// stage 1: ARGMAX_BLOCKS workgroups, each over a contiguous slice; stage 2: one wave combines and advances.
#define ARGMAX_BLOCKS 128
__device__ __forceinline__ void argmax_combine(float& cv, int& ci, float ov, int oi) {
  // keep the later index among equal maxima; an index of -1 means "empty"
  bool take = oi >= 0 && (ci < 0 || ov > cv || (!(cv > ov) && oi > ci));
  if (take) {
    cv = ov;
    ci = oi;
  }
}
__global__ __launch_bounds__(256) void k_argmax_partial(const float* __restrict__ logits, int n, float* __restrict__ pv,
                                                        int* __restrict__ pi) {
  __shared__ float sv[4];
  __shared__ int si[4];
  const int per = (n + gridDim.x - 1) / gridDim.x;
  const int lo = blockIdx.x * per, hi = min(n, lo + per);
  float bv = -INFINITY;
  int bi = -1;
  for (int i = lo + threadIdx.x; i < hi; i += blockDim.x) argmax_combine(bv, bi, logits[i], i);
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    float ov = __shfl_xor(bv, o, 64);
    int oi = __shfl_xor(bi, o, 64);
    argmax_combine(bv, bi, ov, oi);
  }
  if ((threadIdx.x & 63) == 0) {
    sv[threadIdx.x >> 6] = bv;
    si[threadIdx.x >> 6] = bi;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    for (int w = 1; w < 4; w++) argmax_combine(bv, bi, sv[w], si[w]);
    pv[blockIdx.x] = bv;
    pi[blockIdx.x] = bi;
  }
}
__global__ __launch_bounds__(64) void k_argmax_step(const float* __restrict__ pv, const int* __restrict__ pi, int nparts,
                                                    int* __restrict__ token_d, int* __restrict__ pos_d,
                                                    int* __restrict__ step_d, unsigned* __restrict__ out_tokens, int out_cap,
                                                    int* __restrict__ serial_d) {
  float bv = -INFINITY;
  int bi = -1;
  for (int i = threadIdx.x; i < nparts; i += 64) argmax_combine(bv, bi, pv[i], pi[i]);
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    float ov = __shfl_xor(bv, o, 64);
    int oi = __shfl_xor(bi, o, 64);
    argmax_combine(bv, bi, ov, oi);
  }
  if (threadIdx.x == 0) {
    *token_d = bi;
    int st = *step_d;
    if (st < out_cap) out_tokens[st] = (unsigned)bi;
    *step_d = st + 1;
    *pos_d = *pos_d + 1;
    *serial_d = *serial_d + 1;
  }
}

}  // namespace crabml_hip


// ==============================================================================================================
// Host side: the decode step as a list of segments.  With tensor parallelism (tp_size > 1) every segment ends
// in a partial-sum vector that is all-reduced across ranks (RCCL over xGMI; 2 x dim f32 per layer):
//   segment 2l   : [embed] attn-norm(+ pending residual) -> qkv(local heads) -> attention -> wo(local k-slice)
//   segment 2l+1 : ffn-norm(+ pending residual) -> gate/up(local rows) -> down(local k-slice)
//   segment 2L   : final norm(+ pending residual) -> classifier -> greedy argmax / advance
// Column-parallel: wq/wk/wv by heads, gate/up by rows.  Row-parallel: wo, ffn_down by k.
// ==============================================================================================================
#include <dlfcn.h>

using namespace crabml_hip;

// ---- RCCL, bound at run time (the single-GPU product path never needs it) ------------------------------------
struct crabml_hip_tp_comm {
  crabml_hip_device* dev = nullptr;
  void* nccl = nullptr;  // ncclComm_t
  int nranks = 1, rank = 0;
};
namespace {
struct NcclId {  // ncclUniqueId: 128 opaque bytes, passed BY VALUE to ncclCommInitRank
  char b[128];
};
struct Rccl {
  typedef NcclId IdT;
  void* lib = nullptr;
  int (*GetUniqueId)(void*) = nullptr;
  int (*CommInitRank)(void**, int, NcclId, int) = nullptr;
  int (*CommDestroy)(void*) = nullptr;
  int (*AllReduce)(const void*, void*, size_t, int, int, void*, hipStream_t) = nullptr;
  const char* (*GetErrorString)(int) = nullptr;
};
Rccl* rccl() {
  static Rccl r;
  static bool tried = false;
  if (!tried) {
    tried = true;
    for (const char* name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) {
      r.lib = dlopen(name, RTLD_NOW | RTLD_GLOBAL);
      if (r.lib) break;
    }
    if (r.lib) {
      r.GetUniqueId = (decltype(r.GetUniqueId))dlsym(r.lib, "ncclGetUniqueId");
      r.CommInitRank = (decltype(r.CommInitRank))dlsym(r.lib, "ncclCommInitRank");
      r.CommDestroy = (decltype(r.CommDestroy))dlsym(r.lib, "ncclCommDestroy");
      r.AllReduce = (decltype(r.AllReduce))dlsym(r.lib, "ncclAllReduce");
      r.GetErrorString = (decltype(r.GetErrorString))dlsym(r.lib, "ncclGetErrorString");
    }
  }
  return (r.lib && r.GetUniqueId && r.CommInitRank && r.CommDestroy && r.AllReduce) ? &r : nullptr;
}
}  // namespace

struct crabml_hip_llama {
  crabml_hip_device* dev = nullptr;
  crabml_hip_llama_config_t cfg{};
  uint32_t wtype = 0;
  int tp = 1, tp_rank = 0;
  crabml_hip_tp_comm* comm = nullptr;
  // local (per-rank) geometry
  int hd = 0, npairs = 0, n_heads_l = 0, n_kv_l = 0, dim_l = 0, kv_dim_l = 0, hidden_l = 0;
  std::vector<crabml_hip_buf*> held;  // retained weight buffers
  crabml_hip_buf* token_embed = nullptr;
  crabml_hip_buf* rms_final = nullptr;
  crabml_hip_buf* output = nullptr;
  std::vector<crabml_hip_buf*> rms_att, rms_ffn, wq, wk, wv, wo, gate, down, up;
  // device state
  std::vector<void*> kc, vc;
  size_t kv_bytes = 0;
  float* x = nullptr;        // residual stream (dim), replicated on every rank
  float* partial = nullptr;  // tp > 1: this rank's wo / ffn_down partial sums (dim), all-reduced in place
  float* qbuf = nullptr;     // roped, scaled q (dim_l)
  float* attn = nullptr;     // attention output (dim_l)
  float* h = nullptr;        // ffn hidden (hidden_l), strict mode only
  float* logits = nullptr;   // vocab
  float* tmp = nullptr;      // strict-mode GEMV outputs
  char* act_dim = nullptr;   // Q8_0 planes of the normalized residual (dim)
  char* act_attn = nullptr;  // Q8_0 planes of the attention output (dim_l)
  char* act_hid = nullptr;   // Q8_0 planes of the ffn hidden vector (hidden_l)
  float* rope = nullptr;     // [seq_len][npairs][2]
  int* state = nullptr;      // token, pos, step, sink, serial (never reset), fault
  unsigned long long* slots = nullptr;  // dim/32 {chunk sum, epoch} granules of the norm epilogue
  unsigned long long* hgran = nullptr;  // fused FFN: hidden/4 quant granules + hidden/32 scale granules of h
  bool ffn_fused = false;               // gate/up + ffn_down as one launch (k_ffn)
  bool tp_dry = false;       // CRABML_HIP_LLAMA_TP_DRY_RUN: a lone rank that skips the all-reduces (timing only)
  bool kfused = false;       // Q4_K layers, fast mode: fused GEMV kernels with the Q4_K inner loop (enqueue_segment_k)
  bool generic = false;      // per-op launches (strict-order device, or a weight format without fused kernels)
  uint32_t qt = 0, out_qt = 0;  // vec_dot_rhs_dtype of the layer weights / of the classifier
  float* xn = nullptr;       // generic path: normalized residual (f32, dim)
  bool norm_epi = false;     // fast mode, tp == 1: RMSNorm + quantize run in the wo / ffn_down epilogue
  bool norm_epi_k = false;   // the same for Q4_K layers (Q8_K planes out of the epilogue)
  unsigned* out_tokens = nullptr;
  int out_cap = 0;
  float* am_val = nullptr;  // argmax partials
  int* am_idx = nullptr;
  size_t kv_len = 0;
  // [0]: one attention workgroup per head; [1]: the long-context attention kernels (from attn_long_from positions)
  hipGraph_t graph[2] = {nullptr, nullptr};
  hipGraphExec_t exec[2] = {nullptr, nullptr};
  bool use_graph = false;
  bool capturing = false;
  int attn_variant = 0;         // which of the two the next enqueue emits
  bool attn_long_ok = false;    // f16 cache, head_dim % 32 == 0, group size in {1, 2, 4, 8}, seq_len % 8 == 0
  size_t attn_long_from = 0;    // cached positions (pos + 1) from which variant 1 is used
  float* scores_g = nullptr;    // [n_heads_l][seq_len] f32
  unsigned short* p16 = nullptr;  // [n_heads_l][seq_len] f16 probabilities
  // batched prefill (crabml_hip_llama_prefill): row buffers for pf_cap prompt rows, allocated on first use
  size_t pf_cap = 0;
  int* pf_tokens = nullptr;
  float *pf_x = nullptr, *pf_xn = nullptr, *pf_q = nullptr, *pf_k = nullptr, *pf_v = nullptr, *pf_qr = nullptr, *pf_attn = nullptr,
        *pf_tmp = nullptr, *pf_g = nullptr, *pf_u = nullptr;
  char *pf_act_dim = nullptr, *pf_act_hid = nullptr;
  std::vector<std::pair<void*, size_t>> allocs;
};

namespace {

int dalloc(crabml_hip_llama* c, size_t bytes, void** out) {
  size_t cap = 0;
  CH_TRY(pool_alloc(c->dev, bytes, out, &cap));
  c->allocs.push_back({*out, cap});
  return 0;
}

Planes planes_of(const crabml_hip_buf* b) {
  return Planes{(const i32x4*)b->ptr, (const unsigned short*)((const char*)b->ptr + b->wl.off_scale)};
}

// the planes of a 32-block activation (Q8_0: q | d | isum i32;  Q8_1: q | d | s f16) as the kernels write them
struct ActPtrs {
  signed char* q;
  unsigned short* d;
  void* isum;  // the format's third plane
};
ActPtrs act_ptrs(char* p, size_t n, uint32_t qt) {
  ActLayout al = act_layout(qt, n);
  return ActPtrs{(signed char*)p, (unsigned short*)(p + al.off_d), (void*)(p + al.off_aux)};
}
template <int FMT>
typename ActOf<FMT>::type act_view(const ActPtrs& a) {
  if constexpr (FMT == CRABML_HIP_Q4_1)
    return ActQ8_1{(const i32x4*)a.q, a.d, (const unsigned short*)a.isum};
  else
    return ActQ8_0{(const i32x4*)a.q, a.d, (const int*)a.isum};
}

int n_segments(const crabml_hip_llama* c) { return 2 * (int)c->cfg.n_layers + 1; }

// attention of layer l (llama2.rs:571-590): qbuf x KV cache -> attn (f32), plus its Q8_0 planes for wo when xq != NULL.
// Emits the variant selected in c->attn_variant (0: one workgroup per head, 1: the long-context kernels).
template <int G>
void launch_attn_long(crabml_hip_llama* c, int l, signed char* xq, unsigned short* xd, void* xisum, bool prof) {
  crabml_hip_device* dev = c->dev;
  hipStream_t st = dev->stream;
  const int hd = c->hd, seq_cap = (int)c->cfg.seq_len, n_kv = c->n_kv_l;
  const int* pos_d = c->state + 1;
  const int ts = 256 / G, nsplit = (seq_cap + ts - 1) / ts;
  crabml_hip_device::ProfRec r[3];
  for (int i = 0; i < 3; i++)
    if (prof) prof_begin(dev, &r[i], CRABML_HIP_F32, 7 + i, 0.0);  // stages 7 / 8 / 9: scores / softmax / pv
  launch_k(st, prof ? &r[0] : nullptr, k_attn_scores<G>, dim3(n_kv * nsplit), dim3(256), (size_t)G * hd * sizeof(float),
           (const float*)c->qbuf, (const unsigned short*)c->kc[l], pos_d, c->scores_g, n_kv, hd, seq_cap, nsplit);
  launch_k(st, prof ? &r[1] : nullptr, k_attn_softmax, dim3(c->n_heads_l), dim3(256), (size_t)seq_cap * sizeof(float),
           (const float*)c->scores_g, pos_d, (const unsigned short*)dev->exp_table, c->p16, seq_cap);
  launch_k(st, prof ? &r[2] : nullptr, k_attn_pv<G>, dim3(n_kv * (hd / 32)), dim3(256), 0, (const unsigned short*)c->p16,
           (const unsigned short*)c->vc[l], pos_d, c->attn, xq, xd, xisum, hd, seq_cap, c->qt == CRABML_HIP_Q8_1 ? 1 : 0);
  for (int i = 0; i < 3; i++)
    if (prof) prof_end(dev, &r[i]);
}

void enqueue_attention(crabml_hip_llama* c, int l, signed char* xq, unsigned short* xd, void* xisum, const PrefetchPlan& pf,
                       int spare, bool prof) {
  crabml_hip_device* dev = c->dev;
  hipStream_t st = dev->stream;
  const int hd = c->hd, seq_cap = (int)c->cfg.seq_len, n_heads = c->n_heads_l, n_kv = c->n_kv_l;
  const int* pos_d = c->state + 1;
  if (c->attn_variant == 1) {
    switch (n_heads / n_kv) {
      case 1: launch_attn_long<1>(c, l, xq, xd, xisum, prof); break;
      case 2: launch_attn_long<2>(c, l, xq, xd, xisum, prof); break;
      case 4: launch_attn_long<4>(c, l, xq, xd, xisum, prof); break;
      default: launch_attn_long<8>(c, l, xq, xd, xisum, prof); break;
    }
    return;
  }
  const size_t attn_lds = (size_t)(seq_cap + hd) * sizeof(float);
  crabml_hip_device::ProfRec ar{};
  crabml_hip_device::ProfRec* AR = prof ? &ar : nullptr;
  if (prof) prof_begin(dev, &ar, CRABML_HIP_F32, 7, 0.0);
  if (c->cfg.use_f16_kv_cache)
    launch_k(st, AR, k_attn<true>, dim3(n_heads + spare), dim3(256), attn_lds, (const float*)c->qbuf, (const void*)c->kc[l],
             (const void*)c->vc[l], pos_d, (const unsigned short*)dev->exp_table, c->attn, xq, xd, xisum, n_heads, n_kv, hd,
             seq_cap, pf, c->qt == CRABML_HIP_Q8_1 ? 1 : 0);
  else
    launch_k(st, AR, k_attn<false>, dim3(n_heads + spare), dim3(256), attn_lds, (const float*)c->qbuf, (const void*)c->kc[l],
             (const void*)c->vc[l], pos_d, (const unsigned short*)dev->exp_table, c->attn, xq, xd, xisum, n_heads, n_kv, hd,
             seq_cap, pf, c->qt == CRABML_HIP_Q8_1 ? 1 : 0);
  if (prof) prof_end(dev, &ar);
}

// enqueue segment `seg` of one decode step on the device stream (see the banner above): the fused kernels
// (fast mode, Q4_0 / Q8_0 weights)
template <int FMT>
int enqueue_segment_t(crabml_hip_llama* c, int seg) {
  crabml_hip_device* dev = c->dev;
  hipStream_t st = dev->stream;
  const auto& g = c->cfg;
  const int dim = (int)g.embedding_dim, hd = c->hd, seq_cap = (int)g.seq_len;
  const int dim_l = c->dim_l, kv_dim_l = c->kv_dim_l, hidden_l = c->hidden_l;
  const int n_heads_l = c->n_heads_l;
  const bool kv16 = g.use_f16_kv_cache != 0;
  const bool tp = c->tp > 1;
  const int L = (int)g.n_layers;
  int* token_d = c->state;
  int* pos_d = c->state + 1;
  int* step_d = c->state + 2;
  constexpr bool Q81 = FMT == CRABML_HIP_Q4_1;
  const uint32_t qt = c->qt;
  ActPtrs ad = act_ptrs(c->act_dim, dim, qt), aa = act_ptrs(c->act_attn, dim_l, qt), ah = act_ptrs(c->act_hid, hidden_l, qt);
  // measurement hook: only meaningful for eager launches (events cannot live inside the captured graph)
  const bool prof = dev->prof_on && !c->use_graph && !c->capturing;
  const double blk_b = (double)block_bytes(c->wtype) / 32.0;  // weight bytes per element
  auto P0 = [&](crabml_hip_device::ProfRec* r, uint32_t stage, double rows, double k) {
    return prof ? prof_begin(dev, r, c->wtype, stage, rows * k * blk_b + 4.0 * k + 4.0 * rows) : 0;
  };
  auto P1 = [&](crabml_hip_device::ProfRec* r) { return prof ? prof_end(dev, r) : 0; };
  crabml_hip_device::ProfRec pr{};
  crabml_hip_device::ProfRec* R = prof ? &pr : nullptr;
  const bool do_pf = !(g.flags & CRABML_HIP_LLAMA_NO_PREFETCH);
  auto plan = [&](const crabml_hip_buf* a, const crabml_hip_buf* b, const crabml_hip_buf* cc) {
    PrefetchPlan pf{};
    const crabml_hip_buf* v[3] = {a, b, cc};
    for (int i = 0; i < 3; i++) {
      pf.p[i] = do_pf && v[i] ? v[i]->ptr : nullptr;
      pf.n[i] = do_pf && v[i] ? (v[i]->wl.total / 16) * 16 : 0;
    }
    pf.sink = c->state + 3;
    return pf;
  };
  const int spare = do_pf ? (dev->n_cu > 1 ? dev->n_cu - 1 : 0) : 0;
  const size_t norm_lds = norm_lds_bytes(dim);
  // rmsnorm * weight -> act_dim; with tp the previous segment's all-reduced output is folded into x first
  auto norm_quant = [&](const float* wn, float eps, bool add_pending, const PrefetchPlan& pf) {
    crabml_hip_device::ProfRec nr{};
    if (prof) prof_begin(dev, &nr, CRABML_HIP_F32, 6, 8.0 * dim);
    const float* addv = add_pending ? c->partial : nullptr;
    if (dim <= 4096)
      launch_k(st, prof ? &nr : nullptr, k_norm_quant<4, Q81>, dim3(1 + spare), dim3(1024), norm_lds, c->x, addv, wn, dim, eps, ad.q, ad.d, ad.isum, pf, 1);
    else
      launch_k(st, prof ? &nr : nullptr, k_norm_quant<12, Q81>, dim3(1 + spare), dim3(1024), norm_lds, c->x, addv, wn, dim, eps, ad.q, ad.d, ad.isum, pf, 1);
    if (prof) prof_end(dev, &nr);
  };
  // W(dim x k_local) . act -> x (+= residual) or partial (tp)
  const bool norm_epi = c->norm_epi;
  // wnext / eps_next: the RMSNorm that consumes this GEMV's output (norm epilogue only)
  auto gemv_out = [&](const crabml_hip_buf* w, const ActPtrs& a, int k, uint32_t stage, const float* wnext, float eps_next) -> int {
    CH_TRY(P0(&pr, stage, dim, k));
    float* dst = tp ? c->partial : c->x;
    if (norm_epi) {
      NormGather ng{c->slots, c->slots + dim / 16, c->state + 4, c->state + 5, n_segments(c), seg};
      // long rows (ffn_down): two workgroups per chunk, so that every CU streams (a CU sustains ~26 GB/s here)
      const int split = (g.flags & CRABML_HIP_LLAMA_SPLIT_CHUNKS_ALWAYS)  ? 2
                        : (g.flags & CRABML_HIP_LLAMA_SPLIT_CHUNKS_NEVER) ? 1
                        : (k / 32 >= 256 && dim / 32 <= dev->n_cu)        ? 2
                                                                          : 1;
      if (split == 2)
        launch_k(st, R, k_gemv_res_nq<FMT, 2>, dim3(dim / 16), dim3(1024), 0, planes_of(w), act_view<FMT>(a), (const float*)nullptr, c->x, wnext,
                 eps_next,
                 ad.q, ad.d, ad.isum, ng, k / 32, Planes6{nullptr, 0});
      else
        launch_k(st, R, k_gemv_res_nq<FMT, 1>, dim3(dim / 32), dim3(1024), 0, planes_of(w), act_view<FMT>(a), (const float*)nullptr, c->x, wnext,
                 eps_next,
                 ad.q, ad.d, ad.isum, ng, k / 32, Planes6{nullptr, 0});
    } else if (tp) {
      launch_k(st, R, k_gemv_res<FMT, 1, false>, dim3((dim + 1) / 2), dim3(128), 0, planes_of(w), act_view<FMT>(a), dst, dim, k / 32);
    } else {
      launch_k(st, R, k_gemv_res<FMT, 1, true>, dim3((dim + 1) / 2), dim3(128), 0, planes_of(w), act_view<FMT>(a), dst, dim, k / 32);
    }
    CH_TRY(P1(&pr));
    return 0;
  };

  if (seg == 2 * L) {  // final rmsnorm + classifier (llama2.rs:274-278, 199-208) + greedy sampler
    const void* cls_act = c->act_dim;
    if (c->out_qt != qt) {
      // the classifier has its own rhs type (e.g. Q6_K -> Q8_K): normalize the final x to f32 and quantize for it (the
      // planes the last ffn_down epilogue wrote are in the layers' type and stay unused)
      const size_t nlds = norm_lds_bytes(dim);
      const float* addv = tp ? c->partial : nullptr;
      if (dim <= 4096)
        k_norm_f32<4><<<1, 1024, nlds, st>>>(c->x, addv, (const float*)c->rms_final->ptr, dim, g.rms_norm_eps, c->xn, 1);
      else
        k_norm_f32<12><<<1, 1024, nlds, st>>>(c->x, addv, (const float*)c->rms_final->ptr, dim, g.rms_norm_eps, c->xn, 1);
      if (c->out_qt == CRABML_HIP_F32) {
        cls_act = c->xn;
      } else {
        launch_quantize_act(st, c->out_qt, c->xn, (size_t)dim, c->act_dim);
      }
    } else if (!norm_epi) {
      norm_quant((const float*)c->rms_final->ptr, g.rms_norm_eps, tp, plan(nullptr, nullptr, nullptr));
    }
    if (prof)
      CH_TRY(prof_begin(dev, &pr, c->output->dtype, 5,
                        (double)g.vocab_size * (double)(dim / block_elems(c->output->dtype)) * (double)block_bytes(c->output->dtype) +
                            4.0 * dim + 4.0 * g.vocab_size));
    CH_TRY(launch_gemv(dev, c->output, g.vocab_size, dim, cls_act, 1, c->logits, R));
    CH_TRY(P1(&pr));
    k_argmax_partial<<<ARGMAX_BLOCKS, 256, 0, st>>>(c->logits, (int)g.vocab_size, c->am_val, c->am_idx);
    k_argmax_step<<<1, 64, 0, st>>>(c->am_val, c->am_idx, ARGMAX_BLOCKS, token_d, pos_d, step_d, c->out_tokens, c->out_cap,
                                    c->state + 4);
    CH_HIP(dev, hipGetLastError());
    return 0;
  }
  const int l = seg / 2;
  if ((seg & 1) == 0) {
    if (l == 0)
      k_embed<<<(dim + 255) / 256, 256, 0, st>>>((const char*)c->token_embed->ptr, (int)c->token_embed->dtype,
                                                  c->token_embed->wl.off_scale, token_d, dim, c->x);
    // attention rmsnorm (llama2.rs:230-234)
    if (!norm_epi || l == 0)
      norm_quant((const float*)c->rms_att[l]->ptr, g.rms_norm_eps, tp && l > 0, plan(c->wq[l], c->wk[l], c->wv[l]));
    // q, k, v + rope + scale + KV append (llama2.rs:244-256, 542-554, 561-565), local heads only
    QkvEpi e{c->qbuf, c->kc[l], c->vc[l], c->rope, pos_d, 1.0f / std::sqrt((float)hd), dim_l, kv_dim_l, hd,
             (int)g.rope_dim, c->npairs, seq_cap, kv16 ? 1 : 0};
    const int total_rows = dim_l + 2 * kv_dim_l;
    CH_TRY(P0(&pr, 1, total_rows, dim));
    launch_k(st, R, k_qkv<FMT>, dim3((total_rows / 2 + 1) / 2), dim3(128), 0, planes_of(c->wq[l]), planes_of(c->wk[l]),
             planes_of(c->wv[l]), act_view<FMT>(ad), dim / 32, e, Planes6{nullptr, 0});
    CH_TRY(P1(&pr));
    // attention (llama2.rs:571-590) -> attn (f32) [+ Q8_0 planes for wo]; spare CUs prefetch wo
    const bool attn_quant = (hd % 32) == 0;
    const int attn_spare = do_pf && dev->n_cu > n_heads_l ? dev->n_cu - n_heads_l : 0;
    enqueue_attention(c, l, attn_quant ? aa.q : (signed char*)nullptr, aa.d, aa.isum, plan(c->wo[l], nullptr, nullptr), attn_spare, prof);
    if (!attn_quant) launch_quantize_act(st, qt, c->attn, (size_t)dim_l, c->act_attn);
    // wo (+ residual, llama2.rs:600, 266): k = the local heads' slice
    CH_TRY(gemv_out(c->wo[l], aa, dim_l, 2, (const float*)c->rms_ffn[l]->ptr, 1e-5f));
  } else {
    // ffn rmsnorm, eps = the literal 1e-5 (llama2.rs:611)
    if (!norm_epi) norm_quant((const float*)c->rms_ffn[l]->ptr, 1e-5f, tp, plan(nullptr, nullptr, nullptr));
    const float* wnext_down = (const float*)(l + 1 < L ? c->rms_att[l + 1] : c->rms_final)->ptr;
    const int split_down = (g.flags & CRABML_HIP_LLAMA_SPLIT_CHUNKS_ALWAYS)  ? 2
                           : (g.flags & CRABML_HIP_LLAMA_SPLIT_CHUNKS_NEVER) ? 1
                           : (hidden_l / 32 >= 256 && dim / 32 <= dev->n_cu) ? 2
                                                                             : 1;
    if (c->ffn_fused && norm_epi && split_down == 2 && hidden_l / 32 <= 2 * (dim / 16)) {
      // gate / up + silu * mul + quantize + down + residual + the next rmsnorm / quantize: one launch (k_ffn)
      NormGather ng{c->slots, c->slots + dim / 16, c->state + 4, c->state + 5, n_segments(c), seg};
      HGather hg{c->hgran, c->hgran + hidden_l / 4};
      const ActLayout alh = act_layout(qt, (size_t)hidden_l);
      if (prof)
        CH_TRY(prof_begin(dev, &pr, c->wtype, 10,
                          3.0 * hidden_l * (double)dim * blk_b + 4.0 * dim + 4.0 * (2.0 * hidden_l) + 4.0 * hidden_l + 4.0 * dim));
      launch_k(st, R, k_ffn<FMT>, dim3(dim / 16), dim3(1024), alh.total, planes_of(c->gate[l]), planes_of(c->up[l]),
               planes_of(c->down[l]), act_view<FMT>(ad), (const unsigned short*)dev->exp_table, c->x, wnext_down, g.rms_norm_eps, ad.q,
               (void*)ad.d, ad.isum, ng, hg, dim / 32, hidden_l / 32, (int)alh.off_d, (int)alh.off_aux);
      CH_TRY(P1(&pr));
    } else {
      // gate / up + silu * mul (llama2.rs:620-630), local rows
      CH_TRY(P0(&pr, 3, 2.0 * hidden_l, dim));
      launch_k(st, R, k_gateup_q<FMT>, dim3(hidden_l / 32), dim3(1024), 0, planes_of(c->gate[l]), planes_of(c->up[l]),
               act_view<FMT>(ad), dev->exp_table, ah.q, ah.d, ah.isum, dim / 32);
      CH_TRY(P1(&pr));
      // down (+ residual, llama2.rs:633-636): k = the local hidden slice
      CH_TRY(gemv_out(c->down[l], ah, hidden_l, 4, wnext_down, g.rms_norm_eps));
    }
  }
  CH_HIP(dev, hipGetLastError());
  return 0;
}

// The same segment out of per-op launches: one GEMV launch per weight matrix (any format matmul_vec supports;
// the rhs is quantized to vec_dot_rhs_dtype(weight), buf/api.rs:142-159) plus small epilogue kernels.  Used by
// the strict-order device (scalar summation order, bit-exact against the oracle for every format) and, in fast
// mode, by the formats without fused kernels (Q4_1, Q4_K, Q8_K, F16, F32).  Still one hipGraph per step.
int enqueue_segment_generic(crabml_hip_llama* c, int seg) {
  crabml_hip_device* dev = c->dev;
  hipStream_t st = dev->stream;
  const auto& g = c->cfg;
  const int dim = (int)g.embedding_dim, hd = c->hd, seq_cap = (int)g.seq_len;
  const int dim_l = c->dim_l, kv_dim_l = c->kv_dim_l, hidden_l = c->hidden_l;
  const bool kv16 = g.use_f16_kv_cache != 0;
  const bool strict = dev->strict_order;
  const bool tp = c->tp > 1;
  const int L = (int)g.n_layers;
  int* token_d = c->state;
  int* pos_d = c->state + 1;
  int* step_d = c->state + 2;
  const bool prof = dev->prof_on && !c->use_graph && !c->capturing && !strict;
  crabml_hip_device::ProfRec pr{};
  auto gemv = [&](const crabml_hip_buf* w, int m, int k, const void* act, float* out, uint32_t stage) -> int {
    if (strict) return launch_gemv_strict(dev, w, m, k, act, 1, out);
    if (prof)
      CH_TRY(prof_begin(dev, &pr, w->dtype, stage,
                        (double)m * (double)(k / block_elems(w->dtype)) * (double)block_bytes(w->dtype) + 4.0 * k + 4.0 * m));
    CH_TRY(launch_gemv(dev, w, m, k, act, 1, out, prof ? &pr : nullptr));
    if (prof) CH_TRY(prof_end(dev, &pr));
    return 0;
  };
  // CpuTensorBuf::quantize for the rhs of matmul_vec: F32 is the vector itself
  auto quant = [&](const float* src, int n, uint32_t qt, char* planes) -> const void* {
    if (qt == CRABML_HIP_F32) return src;
    launch_quantize_act(st, qt, src, (size_t)n, planes);
    return planes;
  };
  const size_t norm_lds = norm_lds_bytes(dim);
  auto norm = [&](const float* wn, float eps, bool add_pending) {
    const float* addv = add_pending ? c->partial : nullptr;
    if (dim <= 4096)
      k_norm_f32<4><<<1, 1024, norm_lds, st>>>(c->x, addv, wn, dim, eps, c->xn, strict ? 0 : 1);
    else
      k_norm_f32<12><<<1, 1024, norm_lds, st>>>(c->x, addv, wn, dim, eps, c->xn, strict ? 0 : 1);
  };
  float* dst = tp ? c->partial : c->x;

  if (seg == 2 * L) {
    norm((const float*)c->rms_final->ptr, g.rms_norm_eps, tp);
    const void* act = quant(c->xn, dim, c->out_qt, c->act_dim);
    CH_TRY(gemv(c->output, (int)g.vocab_size, dim, act, c->logits, 5));
    k_argmax_partial<<<ARGMAX_BLOCKS, 256, 0, st>>>(c->logits, (int)g.vocab_size, c->am_val, c->am_idx);
    k_argmax_step<<<1, 64, 0, st>>>(c->am_val, c->am_idx, ARGMAX_BLOCKS, token_d, pos_d, step_d, c->out_tokens, c->out_cap,
                                    c->state + 4);
    CH_HIP(dev, hipGetLastError());
    return 0;
  }
  const int l = seg / 2;
  if ((seg & 1) == 0) {
    if (l == 0)
      k_embed<<<(dim + 255) / 256, 256, 0, st>>>((const char*)c->token_embed->ptr, (int)c->token_embed->dtype,
                                                  c->token_embed->wl.off_scale, token_d, dim, c->x);
    norm((const float*)c->rms_att[l]->ptr, g.rms_norm_eps, tp && l > 0);
    const void* act = quant(c->xn, dim, c->qt, c->act_dim);
    QkvEpi e{c->qbuf, c->kc[l], c->vc[l], c->rope, pos_d, 1.0f / std::sqrt((float)hd), dim_l, kv_dim_l, hd,
             (int)g.rope_dim, c->npairs, seq_cap, kv16 ? 1 : 0};
    const int total_rows = dim_l + 2 * kv_dim_l;
    CH_TRY(gemv(c->wq[l], dim_l, dim, act, c->tmp, 1));
    CH_TRY(gemv(c->wk[l], kv_dim_l, dim, act, c->tmp + dim_l, 1));
    CH_TRY(gemv(c->wv[l], kv_dim_l, dim, act, c->tmp + dim_l + kv_dim_l, 1));
    k_qkv_epi<<<(total_rows / 2 + 255) / 256, 256, 0, st>>>(c->tmp, e);
    enqueue_attention(c, l, nullptr, nullptr, nullptr, PrefetchPlan{}, 0, prof);
    const void* aact = quant(c->attn, dim_l, c->qt, c->act_attn);
    CH_TRY(gemv(c->wo[l], dim, dim_l, aact, c->tmp, 2));
    k_res_epi<<<(dim + 255) / 256, 256, 0, st>>>(c->tmp, dst, dim, tp ? 0 : 1);
  } else {
    norm((const float*)c->rms_ffn[l]->ptr, 1e-5f, tp);  // llama2.rs:611
    const void* act = quant(c->xn, dim, c->qt, c->act_dim);
    CH_TRY(gemv(c->gate[l], hidden_l, dim, act, c->tmp, 3));
    CH_TRY(gemv(c->up[l], hidden_l, dim, act, c->tmp + hidden_l, 3));
    k_gateup_epi<<<(hidden_l + 255) / 256, 256, 0, st>>>(c->tmp, c->tmp + hidden_l, dev->exp_table, c->h, hidden_l);
    const void* hact = quant(c->h, hidden_l, c->qt, c->act_hid);
    CH_TRY(gemv(c->down[l], dim, hidden_l, hact, c->tmp, 4));
    k_res_epi<<<(dim + 255) / 256, 256, 0, st>>>(c->tmp, dst, dim, tp ? 0 : 1);
  }
  CH_HIP(dev, hipGetLastError());
  return 0;
}

// Q4_K and Q4_1 layers (fast mode): the fused GEMV kernels with the format's inner loop against Q8_K / Q8_1
// activation planes.  The rhs quantizer is its own launch here (a Q8_K super-block spans 256 rows: eight 32-row
// workgroups; Q8_1 keeps the same structure), so a layer is 11 launches instead of the per-op path's 18.
template <int FMT>
int enqueue_segment_k(crabml_hip_llama* c, int seg) {
  constexpr uint32_t QT = FMT == CRABML_HIP_Q4_K ? CRABML_HIP_Q8_K : CRABML_HIP_Q8_1;
  constexpr int BE = FMT == CRABML_HIP_Q4_K ? 256 : 32;  // elements per weight block
  typedef typename ActOf<FMT>::type Act;
  crabml_hip_device* dev = c->dev;
  hipStream_t st = dev->stream;
  const auto& g = c->cfg;
  const int dim = (int)g.embedding_dim, hd = c->hd, seq_cap = (int)g.seq_len;
  const int dim_l = c->dim_l, kv_dim_l = c->kv_dim_l, hidden_l = c->hidden_l;
  const bool kv16 = g.use_f16_kv_cache != 0;
  const bool tp = c->tp > 1;
  const int L = (int)g.n_layers;
  int* token_d = c->state;
  int* pos_d = c->state + 1;
  int* step_d = c->state + 2;
  const bool prof = dev->prof_on && !c->use_graph && !c->capturing;
  crabml_hip_device::ProfRec pr{};
  crabml_hip_device::ProfRec* R = prof ? &pr : nullptr;
  const double blk_b = (double)block_bytes(FMT) / (double)BE;
  auto P0 = [&](uint32_t stage, double rows, double k) {
    return prof ? prof_begin(dev, &pr, FMT, stage, rows * k * blk_b + 4.0 * k + 4.0 * rows) : 0;
  };
  auto P1 = [&]() { return prof ? prof_end(dev, &pr) : 0; };
  auto act_k = [&](char* planes, int n) {
    ActLayout al = act_layout(QT, (size_t)n);
    if constexpr (FMT == CRABML_HIP_Q4_K)
      return ActQ8_K{(const i32x4*)planes, (const float*)(planes + al.off_d), (const short*)(planes + al.off_aux)};
    else
      return ActQ8_1{(const i32x4*)planes, (const unsigned short*)(planes + al.off_d), (const unsigned short*)(planes + al.off_aux)};
  };
  auto planes_k = [&](const crabml_hip_buf* b) {
    return Planes{(const i32x4*)b->ptr, (const unsigned short*)((const char*)b->ptr + b->wl.off_scale)};
  };
  // a Q6_K tensor inside a Q4_K layer (attn_v / ffn_down of the *_K_M mixes): handed to the kernel beside the planes
  auto six = [&](const crabml_hip_buf* b) {
    return b->dtype == CRABML_HIP_Q6_K ? Planes6{(const char*)b->ptr, b->wl.off_scale} : Planes6{nullptr, 0};
  };
  const size_t norm_lds = norm_lds_bytes(dim);
  // rmsnorm * weight -> xn -> Q8_K planes (buf_q8_k.rs:84-131)
  auto norm_quant = [&](const float* wn, float eps, bool add_pending, uint32_t qt) -> const void* {
    const float* addv = add_pending ? c->partial : nullptr;
    if (dim <= 4096)
      k_norm_f32<4><<<1, 1024, norm_lds, st>>>(c->x, addv, wn, dim, eps, c->xn, 1);
    else
      k_norm_f32<12><<<1, 1024, norm_lds, st>>>(c->x, addv, wn, dim, eps, c->xn, 1);
    if (qt == CRABML_HIP_F32) return c->xn;
    launch_quantize_act(st, qt, c->xn, (size_t)dim, c->act_dim);
    return c->act_dim;
  };
  float* dst = tp ? c->partial : c->x;
  const bool nepi = FMT == CRABML_HIP_Q4_K && c->norm_epi_k;
  // wnext / eps_next: the RMSNorm that consumes this GEMV's output (norm epilogue only)
  // the rhs of wo / ffn_down quantized by the consuming kernel itself (no quantizer launch)
  const bool qin = nepi && !(g.flags & CRABML_HIP_LLAMA_NO_RHS_PROLOGUE) && dim_l % 256 == 0 && hidden_l % 256 == 0;
  // wnext / eps_next: the RMSNorm that consumes this GEMV's output (norm epilogue only); xin: the f32 rhs
  auto gemv_out = [&](const crabml_hip_buf* w, const Act& a, const float* xin, int k, uint32_t stage, const float* wnext,
                      float eps_next) -> int {
    CH_TRY(P0(stage, dim, k));
    if constexpr (FMT == CRABML_HIP_Q4_K) {
      if (nepi) {
        NormGather ng{c->slots, c->slots + dim / 16, c->state + 4, c->state + 5, n_segments(c), seg};
        ActLayout al = act_layout(QT, (size_t)dim);
        signed char* oq = (signed char*)c->act_dim;
        void* od = (void*)(c->act_dim + al.off_d);
        void* ob = (void*)(c->act_dim + al.off_aux);
        const int split = (g.flags & CRABML_HIP_LLAMA_SPLIT_CHUNKS_ALWAYS)  ? 2
                          : (g.flags & CRABML_HIP_LLAMA_SPLIT_CHUNKS_NEVER) ? 1
                          : (k / 32 >= 256 && dim / 32 <= dev->n_cu)        ? 2
                                                                            : 1;
        const size_t lds = (size_t)k + (size_t)(k / 256) * 4 + (size_t)(k / 16) * 2;
        if (split == 2 && qin)
          launch_k(st, R, k_gemv_res_nq<FMT, 2, true>, dim3(dim / 16), dim3(1024), lds, planes_k(w), a, xin, c->x, wnext, eps_next, oq,
                   od, ob, ng, k / BE, six(w));
        else if (split == 2)
          launch_k(st, R, k_gemv_res_nq<FMT, 2>, dim3(dim / 16), dim3(1024), 0, planes_k(w), a, xin, c->x, wnext, eps_next, oq, od, ob,
                   ng, k / BE, six(w));
        else if (qin)
          launch_k(st, R, k_gemv_res_nq<FMT, 1, true>, dim3(dim / 32), dim3(1024), lds, planes_k(w), a, xin, c->x, wnext, eps_next, oq,
                   od, ob, ng, k / BE, six(w));
        else
          launch_k(st, R, k_gemv_res_nq<FMT, 1>, dim3(dim / 32), dim3(1024), 0, planes_k(w), a, xin, c->x, wnext, eps_next, oq, od, ob,
                   ng, k / BE, six(w));
        return P1();
      }
    }
    if (tp)
      launch_k(st, R, k_gemv_res<FMT, 1, false>, dim3((dim + 1) / 2), dim3(128), 0, planes_k(w), a, dst, dim, k / BE);
    else
      launch_k(st, R, k_gemv_res<FMT, 1, true>, dim3((dim + 1) / 2), dim3(128), 0, planes_k(w), a, dst, dim, k / BE);
    return P1();
  };

  if (seg == 2 * L) {
    const void* act = nepi ? (const void*)c->act_dim : norm_quant((const float*)c->rms_final->ptr, g.rms_norm_eps, tp, c->out_qt);
    if (prof)
      CH_TRY(prof_begin(dev, &pr, c->output->dtype, 5,
                        (double)g.vocab_size * (double)(dim / block_elems(c->output->dtype)) * (double)block_bytes(c->output->dtype) +
                            4.0 * dim + 4.0 * g.vocab_size));
    CH_TRY(launch_gemv(dev, c->output, g.vocab_size, dim, act, 1, c->logits, R));
    CH_TRY(P1());
    k_argmax_partial<<<ARGMAX_BLOCKS, 256, 0, st>>>(c->logits, (int)g.vocab_size, c->am_val, c->am_idx);
    k_argmax_step<<<1, 64, 0, st>>>(c->am_val, c->am_idx, ARGMAX_BLOCKS, token_d, pos_d, step_d, c->out_tokens, c->out_cap,
                                    c->state + 4);
    CH_HIP(dev, hipGetLastError());
    return 0;
  }
  const int l = seg / 2;
  if ((seg & 1) == 0) {
    if (l == 0)
      k_embed<<<(dim + 255) / 256, 256, 0, st>>>((const char*)c->token_embed->ptr, (int)c->token_embed->dtype,
                                                  c->token_embed->wl.off_scale, token_d, dim, c->x);
    if (!nepi || l == 0) norm_quant((const float*)c->rms_att[l]->ptr, g.rms_norm_eps, tp && l > 0, QT);
    QkvEpi e{c->qbuf, c->kc[l], c->vc[l], c->rope, pos_d, 1.0f / std::sqrt((float)hd), dim_l, kv_dim_l, hd,
             (int)g.rope_dim, c->npairs, seq_cap, kv16 ? 1 : 0};
    const int total_rows = dim_l + 2 * kv_dim_l;
    CH_TRY(P0(1, total_rows, dim));
    launch_k(st, R, k_qkv<FMT>, dim3((total_rows / 2 + 1) / 2), dim3(128), 0, planes_k(c->wq[l]), planes_k(c->wk[l]),
             planes_k(c->wv[l]), act_k(c->act_dim, dim), dim / BE, e, six(c->wv[l]));
    CH_TRY(P1());
    enqueue_attention(c, l, nullptr, nullptr, nullptr, PrefetchPlan{}, 0, prof);
    if (!qin) launch_quantize_act(st, QT, c->attn, (size_t)dim_l, c->act_attn);
    CH_TRY(gemv_out(c->wo[l], act_k(c->act_attn, dim_l), c->attn, dim_l, 2, (const float*)c->rms_ffn[l]->ptr, 1e-5f));
  } else {
    if (!nepi) norm_quant((const float*)c->rms_ffn[l]->ptr, 1e-5f, tp, QT);  // llama2.rs:611
    CH_TRY(P0(3, 2.0 * hidden_l, dim));
    if constexpr (FMT == CRABML_HIP_Q4_K) {
      const size_t lds = (size_t)dim + (size_t)(dim / 256) * 4 + (size_t)(dim / 16) * 2;
      launch_k(st, R, k_gateup_k_lds, dim3((hidden_l + 31) / 32), dim3(1024), lds, planes_k(c->gate[l]), planes_k(c->up[l]),
               act_k(c->act_dim, dim), (const unsigned short*)dev->exp_table, c->h, hidden_l, dim / 256);
    } else {
      launch_k(st, R, k_gateup<FMT>, dim3((hidden_l + 1) / 2), dim3(128), 0, planes_k(c->gate[l]), planes_k(c->up[l]),
               act_k(c->act_dim, dim), (const unsigned short*)dev->exp_table, c->h, hidden_l, dim / BE);
    }
    CH_TRY(P1());
    if (!qin) launch_quantize_act(st, QT, c->h, (size_t)hidden_l, c->act_hid);
    CH_TRY(gemv_out(c->down[l], act_k(c->act_hid, hidden_l), c->h, hidden_l, 4,
                    (const float*)(l + 1 < L ? c->rms_att[l + 1] : c->rms_final)->ptr, g.rms_norm_eps));
  }
  CH_HIP(dev, hipGetLastError());
  return 0;
}

int enqueue_segment(crabml_hip_llama* c, int seg) {
  if (c->kfused)
    return c->wtype == CRABML_HIP_Q4_K ? enqueue_segment_k<CRABML_HIP_Q4_K>(c, seg) : enqueue_segment_k<CRABML_HIP_Q4_1>(c, seg);
  if (c->generic) return enqueue_segment_generic(c, seg);
  return c->wtype == CRABML_HIP_Q4_0   ? enqueue_segment_t<CRABML_HIP_Q4_0>(c, seg)
         : c->wtype == CRABML_HIP_Q8_0 ? enqueue_segment_t<CRABML_HIP_Q8_0>(c, seg)
                                       : enqueue_segment_t<CRABML_HIP_Q4_1>(c, seg);
}

int allreduce(crabml_hip_llama* c) {
  crabml_hip_device* dev = c->dev;
  if (c->tp_dry) return 0;  // timing-only rank: the partial sums are left as they are
  Rccl* r = rccl();
  if (!r || !c->comm || !c->comm->nccl) CH_BAIL(dev, CRABML_HIP_UNEXPECTED, "llama tp: no RCCL communicator");
  int rc = r->AllReduce(c->partial, c->partial, c->cfg.embedding_dim, /*ncclFloat32*/ 7, /*ncclSum*/ 0, c->comm->nccl, dev->stream);
  if (rc != 0) CH_BAIL(dev, CRABML_HIP_UNEXPECTED, "ncclAllReduce failed: %s", r->GetErrorString ? r->GetErrorString(rc) : "?");
  return 0;
}

int enqueue_step(crabml_hip_llama* c) {
  const int n = n_segments(c);
  for (int s = 0; s < n; s++) {
    CH_TRY(enqueue_segment(c, s));
    if (c->tp > 1 && s + 1 < n) CH_TRY(allreduce(c));
  }
  return 0;
}

// one decode step at cache position `pos` (the host tracks it; the kernels read their own copy from device memory)
int run_step(crabml_hip_llama* c, size_t pos) {
  const int variant = c->attn_long_ok && pos + 1 >= c->attn_long_from ? 1 : 0;
  if (c->use_graph && c->exec[variant]) {
    CH_HIP(c->dev, hipGraphLaunch(c->exec[variant], c->dev->stream));
    return 0;
  }
  c->attn_variant = variant;
  return enqueue_step(c);
}


// ---- batched prefill ---------------------------------------------------------------------------------------
// B prompt rows at positions pos0 .. pos0 + B - 1 through every layer as (B, k) matmul_vec calls (launch_gemv: MFMA
// GEMM for Q4_0 / Q8_0 and B >= 16), row-wise rmsnorm / quantize / rope / append, and causal attention (row r sees
// pos0 + r + 1 cached positions).  Per row this is the arithmetic of the per-op segment path.
int prefill_alloc(crabml_hip_llama* c, size_t cap) {
  if (c->pf_cap >= cap) return 0;
  if (c->pf_cap != 0) CH_BAIL(c->dev, CRABML_HIP_UNEXPECTED, "llama prefill: row buffers already sized for %zu rows", c->pf_cap);
  const auto& g = c->cfg;
  const size_t dim = g.embedding_dim, kv_dim = (size_t)c->kv_dim_l, hidden = g.hidden_dim;
  auto A = [&](size_t bytes, void** out) { return dalloc(c, bytes ? bytes : 16, out); };
  auto act_bytes = [](uint32_t t, size_t n) { return t == CRABML_HIP_F32 ? (size_t)16 : act_layout(t, n).total; };
  CH_TRY(A(cap * 4, (void**)&c->pf_tokens));
  CH_TRY(A(cap * dim * 4, (void**)&c->pf_x));
  CH_TRY(A(cap * dim * 4, (void**)&c->pf_xn));
  CH_TRY(A(cap * dim * 4, (void**)&c->pf_q));
  CH_TRY(A(cap * kv_dim * 4, (void**)&c->pf_k));
  CH_TRY(A(cap * kv_dim * 4, (void**)&c->pf_v));
  CH_TRY(A(cap * dim * 4, (void**)&c->pf_qr));
  CH_TRY(A(cap * dim * 4, (void**)&c->pf_attn));
  CH_TRY(A(cap * dim * 4, (void**)&c->pf_tmp));
  CH_TRY(A(cap * hidden * 4, (void**)&c->pf_g));
  CH_TRY(A(cap * hidden * 4, (void**)&c->pf_u));
  CH_TRY(A(cap * act_bytes(c->qt, dim), (void**)&c->pf_act_dim));
  CH_TRY(A(cap * act_bytes(c->qt, hidden), (void**)&c->pf_act_hid));
  c->pf_cap = cap;
  return 0;
}


// the row-tiled causal attention of a prefill pass; false = not covered (the caller launches k_attn per (head, row))
template <bool KV16, int G, int R>
bool launch_attn_tile_t(crabml_hip_llama* c, int l, int B, int pos0) {
  const int hd = c->hd, n_heads = c->n_heads_l, n_kv = c->n_kv_l, seq_cap = (int)c->cfg.seq_len;
  const int sstride = (pos0 + B + 3) & ~3;
  const size_t lds = (size_t)(G * R) * (size_t)(hd + sstride) * sizeof(float);
  if (lds > 64 * 1024) return false;
  k_attn_tile<KV16, G, R><<<dim3(n_kv, (B + R - 1) / R), 256, lds, c->dev->stream>>>(
      c->pf_qr, c->kc[l], c->vc[l], c->state + 6, (const unsigned short*)c->dev->exp_table, c->pf_attn, n_heads, n_kv, hd, seq_cap, B,
      sstride);
  return true;
}
bool launch_attn_tile(crabml_hip_llama* c, int l, int B, int pos0) {
  const int hd = c->hd, g = c->n_heads_l / c->n_kv_l;
  const bool kv16 = c->cfg.use_f16_kv_cache != 0;
  if (c->cfg.flags & CRABML_HIP_LLAMA_NO_TILE_ATTENTION) return false;
  if (pos0 + B > 1024 || hd > (kv16 ? 256 : 128) || hd % (kv16 ? 16 : 4) != 0 || c->n_heads_l % c->n_kv_l != 0) return false;
  if (kv16) {
    switch (g) {
      case 1: return launch_attn_tile_t<true, 1, 4>(c, l, B, pos0);
      case 2: return launch_attn_tile_t<true, 2, 4>(c, l, B, pos0);
      case 4: return launch_attn_tile_t<true, 4, 4>(c, l, B, pos0);
      case 8: return launch_attn_tile_t<true, 8, 2>(c, l, B, pos0);
      default: return false;
    }
  }
  switch (g) {
    case 1: return launch_attn_tile_t<false, 1, 4>(c, l, B, pos0);
    case 2: return launch_attn_tile_t<false, 2, 4>(c, l, B, pos0);
    case 4: return launch_attn_tile_t<false, 4, 4>(c, l, B, pos0);
    case 8: return launch_attn_tile_t<false, 8, 2>(c, l, B, pos0);
    default: return false;
  }
}

int prefill_chunk(crabml_hip_llama* c, const uint32_t* tokens, size_t B, size_t pos0, bool want_logits) {
  crabml_hip_device* dev = c->dev;
  hipStream_t st = dev->stream;
  const auto& g = c->cfg;
  const int dim = (int)g.embedding_dim, kv_dim = c->kv_dim_l, hidden = (int)g.hidden_dim, hd = c->hd, seq_cap = (int)g.seq_len;
  const int n_heads = c->n_heads_l, n_kv = c->n_kv_l, L = (int)g.n_layers;
  const bool kv16 = g.use_f16_kv_cache != 0, strict = dev->strict_order;
  const int half = strict ? 0 : 1;
  const unsigned rows = (unsigned)B;
  {
    std::vector<int> h(B + 1);
    for (size_t i = 0; i < B; i++) h[i] = (int)tokens[i];
    CH_HIP(dev, hipMemcpyAsync(c->pf_tokens, h.data(), B * sizeof(int), hipMemcpyHostToDevice, st));
    const int p0 = (int)pos0;
    CH_HIP(dev, hipMemcpyAsync(c->state + 6, &p0, sizeof(int), hipMemcpyHostToDevice, st));
    CH_HIP(dev, hipStreamSynchronize(st));  // the staging vectors go out of scope
  }
  const int* pos_d = c->state + 6;
  const size_t norm_lds = norm_lds_bytes(dim);
  auto norm_rows = [&](const float* wn, float eps) {
    if (dim <= 4096)
      k_norm_f32_rows<4><<<rows, 1024, norm_lds, st>>>(c->pf_x, wn, dim, eps, c->pf_xn, half);
    else
      k_norm_f32_rows<12><<<rows, 1024, norm_lds, st>>>(c->pf_x, wn, dim, eps, c->pf_xn, half);
  };
  // CpuTensorBuf::quantize for the rhs of matmul_vec (buf/api.rs:142-159): F32 weights take the rows as they are
  auto quant_rows = [&](const float* src, int n, char* planes) -> const void* {
    if (c->qt == CRABML_HIP_F32) return src;
    launch_quantize_act_rows(st, c->qt, src, B, (size_t)n, planes);
    return planes;
  };
  auto gemm = [&](const crabml_hip_buf* w, int m, int k, const void* act, float* out) -> int {
    return strict ? launch_gemv_strict(dev, w, m, k, act, B, out) : launch_gemv(dev, w, m, k, act, B, out, nullptr);
  };
  k_embed<<<dim3((dim + 255) / 256, rows), 256, 0, st>>>((const char*)c->token_embed->ptr, (int)c->token_embed->dtype,
                                                         c->token_embed->wl.off_scale, c->pf_tokens, dim, c->pf_x);
  for (int l = 0; l < L; l++) {
    norm_rows((const float*)c->rms_att[l]->ptr, g.rms_norm_eps);  // llama2.rs:230-234
    const void* a = quant_rows(c->pf_xn, dim, c->pf_act_dim);
    CH_TRY(gemm(c->wq[l], dim, dim, a, c->pf_q));  // llama2.rs:244-246
    CH_TRY(gemm(c->wk[l], kv_dim, dim, a, c->pf_k));
    CH_TRY(gemm(c->wv[l], kv_dim, dim, a, c->pf_v));
    QkvEpi e{c->pf_qr, c->kc[l], c->vc[l], c->rope, pos_d, 1.0f / std::sqrt((float)hd), dim, kv_dim, hd,
             (int)g.rope_dim, c->npairs, seq_cap, kv16 ? 1 : 0};
    const int pairs = (dim + 2 * kv_dim) / 2;
    k_qkv_epi_rows<<<dim3((pairs + 255) / 256, rows), 256, 0, st>>>(c->pf_q, c->pf_k, c->pf_v, e);
    if (!launch_attn_tile(c, l, (int)B, (int)pos0)) {  // long prompts / unusual shapes: one workgroup per (head, row)
      const size_t attn_lds = (size_t)(seq_cap + hd) * sizeof(float);
      if (kv16)
        k_attn<true><<<dim3(n_heads, rows), 256, attn_lds, st>>>(c->pf_qr, c->kc[l], c->vc[l], pos_d, (const unsigned short*)dev->exp_table,
                                                                  c->pf_attn, nullptr, nullptr, nullptr, n_heads, n_kv, hd, seq_cap,
                                                                  PrefetchPlan{}, 0);
      else
        k_attn<false><<<dim3(n_heads, rows), 256, attn_lds, st>>>(c->pf_qr, c->kc[l], c->vc[l], pos_d, (const unsigned short*)dev->exp_table,
                                                                   c->pf_attn, nullptr, nullptr, nullptr, n_heads, n_kv, hd, seq_cap,
                                                                   PrefetchPlan{}, 0);
    }
    a = quant_rows(c->pf_attn, dim, c->pf_act_dim);
    CH_TRY(gemm(c->wo[l], dim, dim, a, c->pf_tmp));  // llama2.rs:600
    k_res_epi<<<(unsigned)(((size_t)B * dim + 255) / 256), 256, 0, st>>>(c->pf_tmp, c->pf_x, (int)(B * dim), 1);  // :266
    norm_rows((const float*)c->rms_ffn[l]->ptr, 1e-5f);  // llama2.rs:611
    a = quant_rows(c->pf_xn, dim, c->pf_act_dim);
    CH_TRY(gemm(c->gate[l], hidden, dim, a, c->pf_g));  // llama2.rs:620-630
    CH_TRY(gemm(c->up[l], hidden, dim, a, c->pf_u));
    k_gateup_epi<<<(unsigned)(((size_t)B * hidden + 255) / 256), 256, 0, st>>>(c->pf_g, c->pf_u, (const unsigned short*)dev->exp_table,
                                                                               c->pf_g, (int)(B * hidden));
    a = quant_rows(c->pf_g, hidden, c->pf_act_hid);
    CH_TRY(gemm(c->down[l], dim, hidden, a, c->pf_tmp));  // llama2.rs:633-636
    k_res_epi<<<(unsigned)(((size_t)B * dim + 255) / 256), 256, 0, st>>>(c->pf_tmp, c->pf_x, (int)(B * dim), 1);
  }
  if (want_logits) {  // final rmsnorm + classifier of the last row only (llama2.rs:274-278, 199-208)
    CH_HIP(dev, hipMemcpyAsync(c->x, c->pf_x + (B - 1) * (size_t)dim, (size_t)dim * 4, hipMemcpyDeviceToDevice, st));
    if (dim <= 4096)
      k_norm_f32<4><<<1, 1024, norm_lds, st>>>(c->x, nullptr, (const float*)c->rms_final->ptr, dim, g.rms_norm_eps, c->xn, half);
    else
      k_norm_f32<12><<<1, 1024, norm_lds, st>>>(c->x, nullptr, (const float*)c->rms_final->ptr, dim, g.rms_norm_eps, c->xn, half);
    const void* act = c->xn;
    if (c->out_qt != CRABML_HIP_F32) {
      launch_quantize_act(st, c->out_qt, c->xn, (size_t)dim, c->act_dim);
      act = c->act_dim;
    }
    CH_TRY(strict ? launch_gemv_strict(dev, c->output, g.vocab_size, dim, act, 1, c->logits)
                  : launch_gemv(dev, c->output, g.vocab_size, dim, act, 1, c->logits, nullptr));
  }
  CH_HIP(dev, hipGetLastError());
  return 0;
}

int set_state(crabml_hip_llama* c, size_t token, size_t pos, int step) {
  int st[3] = {(int)token, (int)pos, step};
  CH_HIP(c->dev, hipMemcpyAsync(c->state, st, sizeof st, hipMemcpyHostToDevice, c->dev->stream));
  return 0;
}

}  // namespace

extern "C" {

// ---- tensor-parallel communicator (RCCL) ------------------------------------------------------------------
int crabml_hip_tp_get_unique_id(void* id128) {
  Rccl* r = rccl();
  if (!r || !id128) return CRABML_HIP_UNEXPECTED;
  return r->GetUniqueId(id128) == 0 ? 0 : CRABML_HIP_UNEXPECTED;
}

int crabml_hip_tp_comm_create(crabml_hip_device_t* dev, const void* id128, int nranks, int rank, crabml_hip_tp_comm_t** out) {
  if (!dev || !id128 || !out || nranks < 1 || rank < 0 || rank >= nranks) return CRABML_HIP_BAD_INPUT;
  *out = nullptr;
  Rccl* r = rccl();
  if (!r) CH_BAIL(dev, CRABML_HIP_UNEXPECTED, "librccl.so could not be loaded");
  (void)hipSetDevice(dev->ordinal);
  Rccl::IdT id;
  memcpy(&id, id128, sizeof id);
  void* comm = nullptr;
  int rc = r->CommInitRank(&comm, nranks, id, rank);
  if (rc != 0) CH_BAIL(dev, CRABML_HIP_UNEXPECTED, "ncclCommInitRank failed: %s", r->GetErrorString ? r->GetErrorString(rc) : "?");
  crabml_hip_tp_comm* c = new crabml_hip_tp_comm();
  c->dev = dev;
  c->nccl = comm;
  c->nranks = nranks;
  c->rank = rank;
  *out = c;
  return 0;
}

int crabml_hip_tp_comm_destroy(crabml_hip_tp_comm_t* comm) {
  if (!comm) return 0;
  Rccl* r = rccl();
  if (r && comm->nccl) r->CommDestroy(comm->nccl);
  delete comm;
  return 0;
}

// in-place sum over ranks of an F32 buffer's first n elements, on the device stream (the collective the decode
// step issues twice per layer); exposed so the RCCL path can be exercised on its own
int crabml_hip_tp_all_reduce(crabml_hip_tp_comm_t* comm, crabml_hip_buf_t* buf, size_t n) {
  if (!comm || !buf) return CRABML_HIP_BAD_INPUT;
  crabml_hip_device* dev = comm->dev;
  if (buf->dtype != CRABML_HIP_F32 || n > buf->n_elems) CH_BAIL(dev, CRABML_HIP_TENSOR_ERROR, "tp_all_reduce: needs an f32 buffer of >= n elements");
  Rccl* r = rccl();
  if (!r) CH_BAIL(dev, CRABML_HIP_UNEXPECTED, "librccl.so could not be loaded");
  int rc = r->AllReduce(buf->ptr, buf->ptr, n, 7, 0, comm->nccl, dev->stream);
  if (rc != 0) CH_BAIL(dev, CRABML_HIP_UNEXPECTED, "ncclAllReduce failed: %s", r->GetErrorString ? r->GetErrorString(rc) : "?");
  touch(buf);
  return 0;
}

int crabml_hip_llama_create(crabml_hip_device_t* dev, const crabml_hip_llama_config_t* cfg,
                            const crabml_hip_llama_weights_t* w, crabml_hip_llama_t** out) {
  if (!dev || !cfg || !w || !out) return CRABML_HIP_BAD_INPUT;
  *out = nullptr;
  const auto& g = *cfg;
  const int tp = g.tp_size > 1 ? g.tp_size : 1;
  if (!g.n_heads || !g.n_kv_heads || !g.n_layers || g.embedding_dim % g.n_heads || g.n_heads % g.n_kv_heads)
    CH_BAIL(dev, CRABML_HIP_BAD_INPUT, "llama: inconsistent head configuration");
  if (tp > 8 || g.tp_rank < 0 || g.tp_rank >= tp || g.n_kv_heads % tp || g.hidden_dim % tp)
    CH_BAIL(dev, CRABML_HIP_BAD_INPUT, "llama: tp_size %d must divide n_kv_heads and hidden_dim (and be <= 8)", tp);
  // the F32 KV cache pairs head h with kv head h % n_kv (the batch_matmul broadcast quirk): those sets are not
  // contiguous head slices, so a GQA model shards by heads only with the F16 cache (h / (n_heads / n_kv))
  if (tp > 1 && !g.use_f16_kv_cache && g.n_heads != g.n_kv_heads)
    CH_BAIL(dev, CRABML_HIP_NOT_IMPLEMENTED, "llama: tensor-parallel GQA needs the f16 kv cache");
  const size_t hd = g.embedding_dim / g.n_heads;
  const size_t n_heads_l = g.n_heads / tp, n_kv_l = g.n_kv_heads / tp;
  const size_t dim_l = n_heads_l * hd, kv_dim_l = n_kv_l * hd, hidden_l = g.hidden_dim / tp;
  if (g.embedding_dim % 32 || hidden_l % 32 || dim_l % 32 || (hd & 1) || hd > 256 || (g.rope_dim & 1) || g.rope_dim > hd || !g.seq_len)
    CH_BAIL(dev, CRABML_HIP_NOT_IMPLEMENTED, "llama fused path: needs dim, local dims % 32 == 0, even head_dim <= 256, even rope_dim");
  if (g.embedding_dim > 12288)  // k_norm_quant keeps the row in 64 KiB of LDS
    CH_BAIL(dev, CRABML_HIP_NOT_IMPLEMENTED, "llama fused path: embedding_dim %zu > 12288", g.embedding_dim);
  if ((g.seq_len + hd) * sizeof(float) > 64 * 1024)
    CH_BAIL(dev, CRABML_HIP_NOT_IMPLEMENTED, "llama fused path: seq_len %zu needs more than 64 KiB of LDS for the score row", g.seq_len);
  if (!w->token_embed || !w->rms_final_weight || !w->wq || !w->wk || !w->wv || !w->wo || !w->ffn_gate_weight ||
      !w->ffn_down_weight || !w->ffn_up_weight || !w->rms_att_weight || !w->rms_ffn_weight)
    CH_BAIL(dev, CRABML_HIP_BAD_INPUT, "llama: missing weights");
  const crabml_hip_buf* outw = w->output_weight ? w->output_weight : w->token_embed;
  const uint32_t wt = w->wq[0]->dtype, out_wt = outw->dtype;
  const uint32_t qt = vec_dot_rhs_dtype(wt), out_qt = vec_dot_rhs_dtype(out_wt);
  if (qt == 0xffffffffu || out_qt == 0xffffffffu)
    CH_BAIL(dev, CRABML_HIP_NOT_IMPLEMENTED, "llama: weight dtype %u / classifier dtype %u has no matmul_vec", wt, out_wt);
  // fused kernels exist for Q4_0 / Q8_0 layers (fast mode); everything else runs the per-op segment path
  // (a classifier of another format -- llama.cpp's "Q4_0" files keep output.weight in Q6_K -- does not take the layers
  // off the fused kernels: the final segment quantizes the normalized row for the classifier's own rhs type)
  bool generic = dev->strict_order || (wt != CRABML_HIP_Q4_0 && wt != CRABML_HIP_Q8_0 && wt != CRABML_HIP_Q4_1);
  const bool out_differs = out_wt != wt;
  {
    const size_t be = block_elems(wt) > block_elems(qt) ? block_elems(wt) : block_elems(qt);
    const size_t obe = block_elems(out_wt) > block_elems(out_qt) ? block_elems(out_wt) : block_elems(out_qt);
    if (g.embedding_dim % be || dim_l % be || hidden_l % be || g.embedding_dim % obe)
      CH_BAIL(dev, CRABML_HIP_NOT_IMPLEMENTED, "llama: dim / local dims are not multiples of the %zu-element blocks of dtype %u", be, wt);
  }
  auto check = [&](const crabml_hip_buf* b, size_t m, size_t k, uint32_t t) {
    return b && b->dtype == t && b->n_elems == m * k && (block_elems(t) == 1 || b->k == k);
  };
  // a layer's matrices may differ in GGML type (llama.cpp's *_K_M files: attn_v / ffn_down in Q6_K on some layers) as
  // long as they share the rhs type (buf/api.rs:142-159: every K-quant takes Q8_K): such a model runs the per-op
  // segments, each GEMV picking its kernel by the tensor's own dtype
  bool mixed = false;
  bool mix_v_down_q6k = true;  // every deviating tensor is an attn_v / ffn_down in Q6_K inside a Q4_K layer (the *_K_M recipe)
  auto check_w = [&](const crabml_hip_buf* b, size_t m, size_t k, bool v_or_down) {
    if (!b) return false;
    if (b->dtype != wt) {
      if (vec_dot_rhs_dtype(b->dtype) != qt || k % block_elems(b->dtype)) return false;
      mixed = true;
      if (!(v_or_down && wt == CRABML_HIP_Q4_K && b->dtype == CRABML_HIP_Q6_K)) mix_v_down_q6k = false;
    }
    return check(b, m, k, b->dtype);
  };
  for (size_t l = 0; l < g.n_layers; l++) {
    if (!check_w(w->wq[l], dim_l, g.embedding_dim, false) || !check_w(w->wk[l], kv_dim_l, g.embedding_dim, false) ||
        !check_w(w->wv[l], kv_dim_l, g.embedding_dim, true) || !check_w(w->wo[l], g.embedding_dim, dim_l, false) ||
        !check_w(w->ffn_gate_weight[l], hidden_l, g.embedding_dim, false) ||
        !check_w(w->ffn_up_weight[l], hidden_l, g.embedding_dim, false) ||
        !check_w(w->ffn_down_weight[l], g.embedding_dim, hidden_l, true) ||
        !check(w->rms_att_weight[l], 1, g.embedding_dim, CRABML_HIP_F32) ||
        !check(w->rms_ffn_weight[l], 1, g.embedding_dim, CRABML_HIP_F32))
      CH_BAIL(dev, CRABML_HIP_NOT_IMPLEMENTED,
              "llama fused path: layer %zu weights have an unexpected shape, or dtypes that do not share one rhs dtype (tp=%d)", l, tp);
  }
  if (!check(outw, g.vocab_size, g.embedding_dim, out_wt) || !check(w->rms_final_weight, 1, g.embedding_dim, CRABML_HIP_F32) ||
      w->token_embed->n_elems != g.vocab_size * g.embedding_dim)
    CH_BAIL(dev, CRABML_HIP_NOT_IMPLEMENTED, "llama fused path: classifier / final norm / embedding dtype or shape");

  (void)hipSetDevice(dev->ordinal);
  crabml_hip_llama* c = new crabml_hip_llama();
  c->dev = dev;
  c->cfg = g;
  c->wtype = wt;
  // the Q4_K fused kernels take a Q6_K attn_v / ffn_down beside the Q4_K planes, but only in the norm-epilogue form
  const bool nepi_k_possible = !dev->strict_order && wt == CRABML_HIP_Q4_K && out_qt == CRABML_HIP_Q8_K && tp == 1 &&
                               !(g.flags & (CRABML_HIP_LLAMA_NO_NORM_EPILOGUE | CRABML_HIP_LLAMA_NO_KQUANT_FUSION)) &&
                               g.embedding_dim % 256 == 0 && (int)(g.embedding_dim / 32) <= dev->n_cu;
  const bool mix_fused = mixed && mix_v_down_q6k && nepi_k_possible;
  generic = generic || (mixed && !mix_fused);
  c->generic = generic;
  // Q4_K always; Q4_1 when it cannot take the 5-kernel path (mixed classifier format) or for the A/B flag
  c->kfused = !dev->strict_order && (!mixed || mix_fused) && !(g.flags & CRABML_HIP_LLAMA_NO_KQUANT_FUSION) &&
              (wt == CRABML_HIP_Q4_K || (wt == CRABML_HIP_Q4_1 && (generic || out_differs || (g.flags & CRABML_HIP_LLAMA_Q4_1_SEGMENTS))));
  c->qt = qt;
  c->out_qt = out_qt;
  c->tp = tp;
  c->tp_rank = g.tp_rank;
  c->comm = (crabml_hip_tp_comm*)g.tp_comm;
  c->tp_dry = tp > 1 && !g.tp_comm && (g.flags & CRABML_HIP_LLAMA_TP_DRY_RUN);
  c->hd = (int)hd;
  c->npairs = (int)(g.rope_dim / 2);
  c->n_heads_l = (int)n_heads_l;
  c->n_kv_l = (int)n_kv_l;
  c->dim_l = (int)dim_l;
  c->kv_dim_l = (int)kv_dim_l;
  c->hidden_l = (int)hidden_l;
  auto hold = [&](const crabml_hip_buf* b) {
    crabml_hip_buf* m = const_cast<crabml_hip_buf*>(b);
    crabml_hip_buf_retain(m);
    c->held.push_back(m);
    return m;
  };
  c->token_embed = hold(w->token_embed);
  c->rms_final = hold(w->rms_final_weight);
  c->output = hold(outw);
  for (size_t l = 0; l < g.n_layers; l++) {
    c->rms_att.push_back(hold(w->rms_att_weight[l]));
    c->rms_ffn.push_back(hold(w->rms_ffn_weight[l]));
    c->wq.push_back(hold(w->wq[l]));
    c->wk.push_back(hold(w->wk[l]));
    c->wv.push_back(hold(w->wv[l]));
    c->wo.push_back(hold(w->wo[l]));
    c->gate.push_back(hold(w->ffn_gate_weight[l]));
    c->down.push_back(hold(w->ffn_down_weight[l]));
    c->up.push_back(hold(w->ffn_up_weight[l]));
  }
  int rc = 0;
  auto A = [&](size_t bytes, void** p) {
    if (rc == 0) rc = dalloc(c, bytes, p);
  };
  const size_t es = g.use_f16_kv_cache ? 2 : 4;
  c->kv_bytes = n_kv_l * g.seq_len * hd * es;
  c->kc.resize(g.n_layers);
  c->vc.resize(g.n_layers);
  for (size_t l = 0; l < g.n_layers; l++) {
    A(c->kv_bytes, &c->kc[l]);
    A(c->kv_bytes, &c->vc[l]);
  }
  A(g.embedding_dim * 4, (void**)&c->x);
  A(g.embedding_dim * 4, (void**)&c->partial);
  A(dim_l * 4, (void**)&c->qbuf);
  A(dim_l * 4, (void**)&c->attn);
  A(hidden_l * 4, (void**)&c->h);
  A(g.vocab_size * 4, (void**)&c->logits);
  size_t tmp_n = dim_l + 2 * kv_dim_l;
  if (2 * hidden_l > tmp_n) tmp_n = 2 * hidden_l;
  if (g.embedding_dim > tmp_n) tmp_n = g.embedding_dim;
  A(tmp_n * 4, (void**)&c->tmp);
  {
    auto act_bytes = [](uint32_t t, size_t n) { return t == CRABML_HIP_F32 ? (size_t)16 : act_layout(t, n).total; };
    size_t a_dim = act_bytes(qt, g.embedding_dim), a_out = act_bytes(out_qt, g.embedding_dim);
    A(a_dim > a_out ? a_dim : a_out, (void**)&c->act_dim);
    A(act_bytes(qt, dim_l), (void**)&c->act_attn);
    A(act_bytes(qt, hidden_l), (void**)&c->act_hid);
    A(g.embedding_dim * 4, (void**)&c->xn);
  }
  A(g.seq_len * (size_t)(c->npairs ? c->npairs : 1) * 2 * 4, (void**)&c->rope);
  {
    const size_t grp = n_heads_l / n_kv_l;
    c->attn_long_ok = g.use_f16_kv_cache && hd % 32 == 0 && g.seq_len % 8 == 0 && (grp == 1 || grp == 2 || grp == 4 || grp == 8) &&
                      !(g.flags & CRABML_HIP_LLAMA_NO_LONG_ATTENTION);
    c->attn_long_from = g.attn_long_from ? g.attn_long_from : 224;  // measured crossover on MI355X (Llama-3-8B shape): ~200-220
    if (c->attn_long_ok) {
      A(n_heads_l * g.seq_len * 4, (void**)&c->scores_g);
      A(n_heads_l * g.seq_len * 2, (void**)&c->p16);
    }
  }
  A(8 * sizeof(int), (void**)&c->state);
  A((g.embedding_dim / 16 + g.embedding_dim) * 8, (void**)&c->slots);
  c->norm_epi = !generic && tp == 1 && !(g.flags & CRABML_HIP_LLAMA_NO_NORM_EPILOGUE) &&
                (int)(g.embedding_dim / 32) <= dev->n_cu;  // every workgroup of the gather must be resident
  c->ffn_fused = c->norm_epi && (g.flags & CRABML_HIP_LLAMA_FFN_FUSION);  // opt-in: measured slower than the two kernels
  if (c->ffn_fused) {
    A((hidden_l / 4 + hidden_l / 32) * 8, (void**)&c->hgran);
  }
  c->norm_epi_k = c->kfused && wt == CRABML_HIP_Q4_K && out_qt == CRABML_HIP_Q8_K && tp == 1 &&
                  !(g.flags & CRABML_HIP_LLAMA_NO_NORM_EPILOGUE) && g.embedding_dim % 256 == 0 && (int)(g.embedding_dim / 32) <= dev->n_cu;
  c->out_cap = (int)g.seq_len;
  A((size_t)c->out_cap * 4, (void**)&c->out_tokens);
  A(ARGMAX_BLOCKS * 4, (void**)&c->am_val);
  A(ARGMAX_BLOCKS * 4, (void**)&c->am_idx);
  if (rc != 0) {
    crabml_hip_llama_destroy(c);
    return rc;
  }
  // RoPE table with the reference's own recurrence (rope.rs:47-54: theta_scale = 10000^(-2/hd), theta = pos,
  // theta *= theta_scale per pair; base hard-coded) evaluated with the host libm, as the trait op does.
  {
    std::vector<float> tab(g.seq_len * (size_t)(c->npairs ? c->npairs : 1) * 2, 0.f);
    const float theta_scale = powf(10000.0f, -2.0f / (float)hd);
    for (size_t p = 0; p < g.seq_len; p++) {
      float theta = (float)p;
      for (int i = 0; i < c->npairs; i++) {
        tab[(p * c->npairs + i) * 2] = cosf(theta);
        tab[(p * c->npairs + i) * 2 + 1] = sinf(theta);
        theta *= theta_scale;
      }
    }
    hipError_t e = hipMemcpyAsync(c->rope, tab.data(), tab.size() * 4, hipMemcpyHostToDevice, dev->stream);
    if (e == hipSuccess) e = hipMemsetAsync(c->state, 0, 8 * sizeof(int), dev->stream);
    if (e == hipSuccess) e = hipMemsetAsync(c->slots, 0, (g.embedding_dim / 16 + g.embedding_dim) * 8, dev->stream);
    if (e == hipSuccess && c->hgran) e = hipMemsetAsync(c->hgran, 0, (hidden_l / 4 + hidden_l / 32) * 8, dev->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(dev->stream);
    if (e != hipSuccess) {
      crabml_hip_llama_destroy(c);
      return hip_fail(dev, e, "llama init", __FILE__, __LINE__);
    }
  }
  // capture one decode step into a graph (token / pos / step are read from device memory by the kernels).
  // tp > 1 without a communicator = a rank of the single-device simulation: driven segment by segment, no graph.
  // tp > 1 over RCCL launches eagerly unless CRABML_HIP_LLAMA_TP_GRAPH asks for the collectives to be captured too.
  const bool want_graph = !(g.flags & CRABML_HIP_LLAMA_NO_GRAPH) &&
                          (tp == 1 || c->tp_dry || (c->comm != nullptr && (g.flags & CRABML_HIP_LLAMA_TP_GRAPH)));
  if (want_graph) {
    const int nvar = c->attn_long_ok ? 2 : 1;
    bool ok = true;
    for (int v = 0; v < nvar && ok; v++) {
      ok = false;
      hipError_t e = hipStreamBeginCapture(dev->stream, hipStreamCaptureModeThreadLocal);
      if (e != hipSuccess) break;
      c->capturing = true;
      c->attn_variant = v;
      int erc = enqueue_step(c);
      c->capturing = false;
      hipGraph_t graph = nullptr;
      hipError_t e2 = hipStreamEndCapture(dev->stream, &graph);
      if (erc == 0 && e2 == hipSuccess && graph) {
        hipGraphExec_t exec = nullptr;
        if (hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0) == hipSuccess) {
          c->graph[v] = graph;
          c->exec[v] = exec;
          ok = true;
        } else {
          (void)hipGraphDestroy(graph);
        }
      } else if (graph) {
        (void)hipGraphDestroy(graph);
      }
    }
    (void)hipGetLastError();
    c->use_graph = ok;
    c->attn_variant = 0;
    if (!ok && tp == 1) {  // fail loudly: the caller asked for the graph path
      crabml_hip_llama_destroy(c);
      CH_BAIL(dev, CRABML_HIP_UNEXPECTED, "llama: hipGraph capture/instantiate failed");
    }
    // tp > 1: if RCCL could not be captured the step simply runs eagerly
  }
  *out = c;
  return 0;
}

int crabml_hip_llama_destroy(crabml_hip_llama_t* c) {
  if (!c) return 0;
  (void)hipStreamSynchronize(c->dev->stream);
  for (int v = 0; v < 2; v++) {
    if (c->exec[v]) (void)hipGraphExecDestroy(c->exec[v]);
    if (c->graph[v]) (void)hipGraphDestroy(c->graph[v]);
  }
  for (auto& a : c->allocs) pool_free(c->dev, a.first, a.second);
  for (auto* b : c->held) crabml_hip_buf_release(b);
  delete c;
  return 0;
}

static int check_step(crabml_hip_llama* c, size_t token, size_t pos) {
  crabml_hip_device* dev = c->dev;
  if (token >= c->cfg.vocab_size) CH_BAIL(dev, CRABML_HIP_BAD_INPUT, "llama: token %zu out of range", token);
  if (pos != c->kv_len) CH_BAIL(dev, CRABML_HIP_BAD_INPUT, "llama: pos %zu != kv cache length %zu", pos, c->kv_len);
  if (pos >= c->cfg.seq_len) CH_BAIL(dev, CRABML_HIP_TENSOR_ERROR, "llama: kv cache is full (%zu)", c->cfg.seq_len);
  return 0;
}

int crabml_hip_llama_forward(crabml_hip_llama_t* c, size_t token, size_t pos, float* logits) {
  if (!c) return CRABML_HIP_BAD_INPUT;
  crabml_hip_device* dev = c->dev;
  if (c->tp > 1 && !c->comm && !c->tp_dry)
    CH_BAIL(dev, CRABML_HIP_BAD_INPUT, "llama: a tp rank without a communicator is driven by crabml_hip_llama_tp_sim_*");
  CH_TRY(check_step(c, token, pos));
  CH_TRY(set_state(c, token, pos, 0));
  CH_TRY(run_step(c, pos));
  c->kv_len++;
  if (logits) {
    int fault = 0;
    CH_HIP(dev, hipMemcpyAsync(logits, c->logits, c->cfg.vocab_size * 4, hipMemcpyDeviceToHost, dev->stream));
    CH_HIP(dev, hipMemcpyAsync(&fault, c->state + 5, sizeof(int), hipMemcpyDeviceToHost, dev->stream));
    CH_HIP(dev, hipStreamSynchronize(dev->stream));
    if (fault) CH_BAIL(dev, CRABML_HIP_UNEXPECTED, "llama: a norm-epilogue gather timed out (workgroups not co-resident?)");
  }
  return 0;
}

int crabml_hip_llama_decode_greedy(crabml_hip_llama_t* c, size_t token, size_t n_steps, uint32_t* out_tokens) {
  if (!c || (!out_tokens && n_steps)) return CRABML_HIP_BAD_INPUT;
  crabml_hip_device* dev = c->dev;
  if (c->tp > 1 && !c->comm && !c->tp_dry)
    CH_BAIL(dev, CRABML_HIP_BAD_INPUT, "llama: a tp rank without a communicator is driven by crabml_hip_llama_tp_sim_*");
  if (token >= c->cfg.vocab_size) CH_BAIL(dev, CRABML_HIP_BAD_INPUT, "llama: token %zu out of range", token);
  if (c->kv_len + n_steps > c->cfg.seq_len || n_steps > (size_t)c->out_cap)
    CH_BAIL(dev, CRABML_HIP_TENSOR_ERROR, "llama: %zu steps do not fit the kv cache (%zu of %zu used)", n_steps, c->kv_len, c->cfg.seq_len);
  if (n_steps == 0) return 0;
  CH_TRY(set_state(c, token, c->kv_len, 0));
  for (size_t s = 0; s < n_steps; s++) CH_TRY(run_step(c, c->kv_len + s));
  c->kv_len += n_steps;
  int fault = 0;
  CH_HIP(dev, hipMemcpyAsync(out_tokens, c->out_tokens, n_steps * 4, hipMemcpyDeviceToHost, dev->stream));
  CH_HIP(dev, hipMemcpyAsync(&fault, c->state + 5, sizeof(int), hipMemcpyDeviceToHost, dev->stream));
  CH_HIP(dev, hipStreamSynchronize(dev->stream));
  if (fault) CH_BAIL(dev, CRABML_HIP_UNEXPECTED, "llama: a norm-epilogue gather timed out (workgroups not co-resident?)");
  return 0;
}


int crabml_hip_llama_prefill(crabml_hip_llama_t* c, const uint32_t* tokens, size_t n, float* logits) {
  if (!c || (!tokens && n)) return CRABML_HIP_BAD_INPUT;
  crabml_hip_device* dev = c->dev;
  if (n == 0) CH_BAIL(dev, CRABML_HIP_BAD_INPUT, "llama prefill: expected at least 1 prompt token");  // llama2.rs:117-122
  for (size_t i = 0; i < n; i++)
    if (tokens[i] >= c->cfg.vocab_size) CH_BAIL(dev, CRABML_HIP_BAD_INPUT, "llama: token %u out of range", tokens[i]);
  if (c->kv_len + n > c->cfg.seq_len)
    CH_BAIL(dev, CRABML_HIP_TENSOR_ERROR, "llama: %zu prompt tokens do not fit the kv cache (%zu of %zu used)", n, c->kv_len, c->cfg.seq_len);
  if (c->tp > 1) {  // token loop
    for (size_t i = 0; i < n; i++) CH_TRY(crabml_hip_llama_forward(c, tokens[i], c->kv_len, i + 1 == n ? logits : nullptr));
    return 0;
  }
  const size_t chunk = c->cfg.prefill_chunk ? c->cfg.prefill_chunk : 512;  // 8B shape: 12.6k / 17.2k prompt tok/s at 256 / 512 rows
  CH_TRY(prefill_alloc(c, chunk));
  for (size_t i = 0; i < n; i += chunk) {
    const size_t B = n - i < chunk ? n - i : chunk;
    CH_TRY(prefill_chunk(c, tokens + i, B, c->kv_len, logits != nullptr && i + B == n));
    c->kv_len += B;
  }
  if (logits) {
    CH_HIP(dev, hipMemcpyAsync(logits, c->logits, c->cfg.vocab_size * 4, hipMemcpyDeviceToHost, dev->stream));
    CH_HIP(dev, hipStreamSynchronize(dev->stream));
  }
  return 0;
}

// Single-device simulation of a tensor-parallel group: `ranks[r]` was created with tp_size = n, tp_rank = r,
// tp_comm = NULL on the SAME device.  Segments are enqueued rank by rank and the all-reduce is a local kernel
// (sum in rank order).  Validates the sharding, the partial-sum plumbing and the residual hand-off on one GPU.
int crabml_hip_llama_tp_sim_forward(crabml_hip_llama_t* const* ranks, int n, size_t token, size_t pos, float* logits) {
  if (!ranks || n < 1 || n > 8 || !ranks[0]) return CRABML_HIP_BAD_INPUT;
  crabml_hip_device* dev = ranks[0]->dev;
  for (int r = 0; r < n; r++) {
    if (!ranks[r] || ranks[r]->dev != dev || ranks[r]->tp != n || ranks[r]->tp_rank != r || ranks[r]->comm)
      CH_BAIL(dev, CRABML_HIP_BAD_INPUT, "tp_sim: rank %d is not a communicator-less rank %d of %d on this device", r, r, n);
    CH_TRY(check_step(ranks[r], token, pos));
    CH_TRY(set_state(ranks[r], token, pos, 0));
  }
  const int nseg = n_segments(ranks[0]);
  SimPtrs ptrs{};
  for (int r = 0; r < n; r++) ptrs.p[r] = ranks[r]->partial;
  const int dim = (int)ranks[0]->cfg.embedding_dim;
  for (int r = 0; r < n; r++) ranks[r]->attn_variant = ranks[r]->attn_long_ok && pos + 1 >= ranks[r]->attn_long_from ? 1 : 0;
  for (int s = 0; s < nseg; s++) {
    for (int r = 0; r < n; r++) CH_TRY(enqueue_segment(ranks[r], s));
    if (n > 1 && s + 1 < nseg) k_sim_allreduce<<<(dim + 255) / 256, 256, 0, dev->stream>>>(ptrs, n, dim);
  }
  for (int r = 0; r < n; r++) ranks[r]->kv_len++;
  if (logits) {
    CH_HIP(dev, hipMemcpyAsync(logits, ranks[0]->logits, ranks[0]->cfg.vocab_size * 4, hipMemcpyDeviceToHost, dev->stream));
    CH_HIP(dev, hipStreamSynchronize(dev->stream));
  }
  return 0;
}

size_t crabml_hip_llama_kv_len(const crabml_hip_llama_t* c) { return c ? c->kv_len : 0; }

int crabml_hip_llama_reset(crabml_hip_llama_t* c) {
  if (!c) return CRABML_HIP_BAD_INPUT;
  c->kv_len = 0;
  return 0;
}

int crabml_hip_llama_debug_kv(crabml_hip_llama_t* c, size_t layer, int32_t which_v, void* dst, size_t nbytes) {
  if (!c || !dst) return CRABML_HIP_BAD_INPUT;
  if (layer >= c->cfg.n_layers || nbytes > c->kv_bytes) CH_BAIL(c->dev, CRABML_HIP_BAD_INPUT, "llama debug_kv: bad layer/size");
  CH_HIP(c->dev, hipMemcpyAsync(dst, which_v ? c->vc[layer] : c->kc[layer], nbytes, hipMemcpyDeviceToHost, c->dev->stream));
  CH_HIP(c->dev, hipStreamSynchronize(c->dev->stream));
  return 0;
}

}  // extern "C"

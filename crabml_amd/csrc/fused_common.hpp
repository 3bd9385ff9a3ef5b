// fused_common.hpp -- rhs quantizer lanes, weight prefetch, embedding, RMSNorm (+ quantize), q/k/v GEMV + rope + KV append, softmax row
// Part of the fused decode step (fused.hip includes the three fused_*.hpp files once, in order; they are not stand-alone
// translation units: the kernels are launched from fused.hip's host code).
#pragma once
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

// ---- Q8_K planes straight from the kernels that produce the vector ------------------------------------------------------
// A Q8_K super-block is 256 elements and its scale comes from the FIRST element of maximal |x| among them
// (buf_q8_k.rs:84-131), but the producers own less: an attention workgroup one head (head_dim values), a gate/up workgroup
// 32 rows.  Every producer publishes its values as 8-byte {f32, epoch} granules (one write-through store carries data and
// tag), gathers the rest of its super-block from its neighbours' granules (bounded polls), runs the whole block's quantizer
// (q8k_wave_quant: the arithmetic of the stand-alone quantizer launch, bit for bit) and stores the part that is its own.
// The consuming GEMV then only copies finished planes into LDS instead of quantizing the f32 vector in its prologue
// (56 super-blocks per workgroup for ffn_down).  All producers of a super-block are co-resident by construction.
struct Q8KExchange {
  unsigned long long* gran;  // one granule per element of the vector
  const int* serial;         // decode-step serial number (never reset): epoch = serial * nseg + seg + 1
  int* fault;
  int nseg, seg;
};
// called by ONE whole wave.  own: this workgroup's n_own consecutive values (LDS), the first of which is element `first` of
// the vector; n_own divides 256 and is a multiple of 16.  oq / od / obs / oqp: the Q8_K planes of the vector (q | d | bsums | qp).
__device__ __forceinline__ void q8k_exchange_store(const Q8KExchange& ex, const float* own, int first, int n_own, int lane,
                                                   signed char* __restrict__ oq, float* __restrict__ od, short* __restrict__ obs,
                                                   signed char* __restrict__ oqp) {
  const unsigned epoch = (unsigned)(*ex.serial) * (unsigned)ex.nseg + (unsigned)ex.seg + 1u;
  if (n_own < 256) {
    for (int i = lane; i < n_own; i += 64)
      __hip_atomic_store(ex.gran + first + i, ((unsigned long long)epoch << 32) | (unsigned long long)__builtin_bit_cast(unsigned, own[i]),
                         __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __builtin_amdgcn_s_waitcnt(0x0F70);  // vmcnt(0): the granules are on their way before the polls queue up behind them
  }
  const int sb = first >> 8, e0 = sb * 256 + 4 * lane;  // this lane's four elements of the super-block
  const bool mine = e0 >= first && e0 < first + n_own;
  f32x4 v;
  if (mine) {
#pragma unroll
    for (int i = 0; i < 4; i++) v[i] = own[e0 - first + i];
  } else {
    unsigned long long g[4];
#pragma unroll
    for (int i = 0; i < 4; i++) g[i] = __hip_atomic_load(ex.gran + e0 + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#pragma unroll
    for (int i = 0; i < 4; i++) {
      int tries = 0;
      while ((unsigned)(g[i] >> 32) != epoch && tries < (1 << 21)) {
        __builtin_amdgcn_s_sleep(2);
        g[i] = __hip_atomic_load(ex.gran + e0 + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        tries++;
      }
      if ((unsigned)(g[i] >> 32) != epoch) *ex.fault = 1;  // a neighbour never arrived: flagged, not hung
      v[i] = __builtin_bit_cast(float, (unsigned)g[i]);
    }
  }
  const Q8KLane o = q8k_wave_quant(v, lane);
  if (mine) {
    ((unsigned*)oq)[sb * 64 + lane] = o.packed;
    q8k_store_class_major(oqp + sb * 256, lane, o.packed);
    if ((lane & 3) == 0) obs[sb * 16 + (lane >> 2)] = (short)o.quad_sum;
  }
  if (lane == 0 && first == sb * 256) od[sb] = o.d;
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
      if (half) {
        // fast mode: 64 chunk sums per round through the DPP tree (wave_sum_f32), rounds added in order -- the order of the
        // wo / ffn_down norm epilogue, where a 128-step dependent v_add chain sat between the last arriving granule and the
        // quantizer (0.4 us per hop on MI355X: a dependent f32 add issues every ~8 cycles)
        sum += wave_sum_f32(v);
      } else {
#pragma unroll
        for (int i = 0; i < 64; i++) sum += rl_f(v, i);
      }
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

// batched prefill, Q8_0 / Q8_1 rhs: x[row] (+= addv[row]: the pending wo / ffn_down output, llama2.rs:266 / :636) -> RMSNorm ->
// the row's quantized planes, one workgroup per row.  Replaces residual-add, norm and quantize launches (three passes over
// the (rows, cols) activations) by one; per row the arithmetic is norm_quant_block's, i.e. the decode step's.
template <int NIT, bool Q81>
__global__ __launch_bounds__(1024) void k_norm_quant_rows(float* __restrict__ x, const float* __restrict__ addv, const float* __restrict__ w,
                                                         int cols, float eps, char* __restrict__ planes, size_t row_stride, size_t off_d,
                                                         size_t off_aux, int half) {
  extern __shared__ float lds[];
  __shared__ float s_rms;
  NormLds L{lds, lds + cols};
  const size_t r = blockIdx.x;
  char* p = planes + r * row_stride;
  norm_quant_block<NIT, true, Q81>(x + r * cols, addv ? addv + r * cols : nullptr, w, cols, eps, L, &s_rms, (signed char*)p,
                                   (unsigned short*)(p + off_d), (void*)(p + off_aux), nullptr, half);
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
// the same without a load inside a branch (the rotation of a clamped pair is fetched unconditionally and ignored where `rot` is
// false): a loaded value that leaves a lane-predicated region is copied at its end, i.e. waited for on the spot -- a memory round
// trip in front of the wave's first weight request
__device__ __forceinline__ QkvPre qkv_preload_nb(const QkvEpi& e, int row0) {
  QkvPre p;
  p.pos = *e.pos_d;
  const int i = (row0 < e.dim ? row0 : row0 - e.dim) % e.hd;
  p.rot = row0 < e.dim + e.kv_dim && i < e.rope_dim;
  const float* cs = e.rope + ((size_t)p.pos * e.npairs + (p.rot ? (i >> 1) : 0)) * 2;
  p.c = cs[0];
  p.s = cs[1];
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

// DEFER: the rhs planes come from a hop-free ffn_down launch -- the row dots are multiplied by 1 / rms (RmsTail, gemv_core.hpp)
template <int FMT, bool DEFER = false>
__global__ __launch_bounds__(128) void k_qkv(Planes wq, Planes wk, Planes wv, typename ActOf<FMT>::type act, int nb, QkvEpi e,
                                             Planes6 wv6, RmsTail rt, int upfront = 0) {
  const int lane = threadIdx.x & 63;
  const int wave = blockIdx.x * (blockDim.x >> 6) + wave_in_wg();
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
  RmsReq rq{0.f, 0.f};
  if constexpr (DEFER) {
    // (every lane loads the pair's rotation -- one address -- instead of lane 0 alone: with the chunk sums requested next to it, a
    // load inside a lane-predicated branch would make the wave wait for all of them before its first weight request)
    pre = qkv_preload_nb(e, row0);
    rq = rms_request(rt, lane);
  } else {
    if (lane == 0) pre = qkv_preload(e, row0);
  }
  float acc[2];
  bool done = false;
  if constexpr (FMT == CRABML_HIP_Q4_K) {
    if (wv6.base != nullptr && row0 >= e.dim + e.kv_dim) {  // the V rows of this layer are Q6_K (wave-uniform)
      rows_partial_q6k<2>(wv6.base, wv6.off_qh, act, local, m, nb, lane, acc);
      done = true;
    }
  }
  float inv_rms = 1.0f;
  if constexpr (DEFER && FMT != CRABML_HIP_Q4_K) {
    if (upfront && nb * BlockFmt<FMT>::UNITS == 128)
      inv_rms = rows_partial_rms_128<FMT, 2>(w.q, w.d, act, local, m, nb, lane, acc, rt, rq);
    else
      inv_rms = rows_partial_rms<FMT, 2>(w.q, w.d, act, local, m, nb, lane, acc, rt, rq);
  } else {
    if constexpr (FMT != CRABML_HIP_Q4_K) {
      if (!done && upfront && (nb * BlockFmt<FMT>::UNITS) % 128 == 0) {
        rows_partial_2step<FMT, 2>(w.q, w.d, act, local, m, nb, lane, acc);
        done = true;
      }
    }
    if (!done) rows_dot<FMT, 2>(w.q, w.d, act, local, m, nb, lane, acc);
  }
  float s0 = wave_sum_f32(acc[0]), s1 = wave_sum_f32(acc[1]);
  if constexpr (DEFER) {
    s0 *= inv_rms;
    s1 *= inv_rms;
  }
  if (lane == 0) qkv_epilogue(e, pre, row0, s0, s1);
}
// strict order (CRABML_HIP_FLAG_STRICT_ORDER, Q4_0 / Q8_0 / Q4_1 layers): the same launch with the block terms parked in LDS and
// added in block order by one lane per row (rows_terms / ordered_sum, gemv_core.hpp) -- q, k and v rows bit-identical to the scalar
// loops of the reference, then the same epilogue.  Workgroup = 4 waves x one (even, odd) row pair; dynamic LDS = 8 * nt floats.
template <int FMT>
__global__ __launch_bounds__(256) void k_qkv_ord(Planes wq, Planes wk, Planes wv, typename ActOf<FMT>::type act, int nb, QkvEpi e, Planes6 wv6) {
  extern __shared__ __attribute__((aligned(16))) float ord_terms[];
  const int lane = threadIdx.x & 63, wv_i = wave_in_wg();
  const int wave = blockIdx.x * (blockDim.x >> 6) + wv_i;
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
  float s = 0.0f;
  if constexpr (FMT == CRABML_HIP_Q4_K) {
    // nb super-blocks, nine terms each (q4k_class_terms / q4k_ordered_sum, gemv_core.hpp); dynamic LDS = 8 * q4k_rec_stride(nb) floats
    const int stride = q4k_rec_stride(nb);
    float* T = ord_terms + (size_t)wv_i * 2 * stride;
    if (wv6.base != nullptr && row0 >= e.dim + e.kv_dim)  // the V rows of this layer are Q6_K (wave-uniform): the same records
      rows_terms_q6k<2>(wv6.base, wv6.off_qh, act, local, m, nb, lane, T, stride);
    else
      rows_terms_q4k<2, true>(w.q, (const i32x4*)w.d, act, local, m, nb, lane, T, stride);
    __builtin_amdgcn_s_waitcnt(0xC07F);  // lgkmcnt(0): this wave's own LDS stores have landed
    __builtin_amdgcn_wave_barrier();
    if (lane < 2) s = q4k_ordered_sum(T + lane * stride, nb);
  } else {
    const int nt = (nb + 3) & ~3;
    float* T = ord_terms + (size_t)wv_i * 2 * nt;
    rows_terms<FMT, 2, typename ActOf<FMT>::type, true>(w.q, w.d, act, local, m, nb, lane, T, nt);
    __builtin_amdgcn_wave_barrier();
    if (lane < 2) s = ordered_sum(T + lane * nt, nb);
  }
  const float s1 = __shfl(s, 1, 64);
  if (lane == 0) qkv_epilogue(e, pre, row0, s, s1);
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

// softmax.rs:36-54 over scores[0..seq) in LDS, in place, by a workgroup of NW waves (4 or 16): max, exp through the f16
// table, row sum sequential up to 1024 positions (bit-exact; at every length when seq_sum is set) and a block tree beyond,
// true division.  F16: the
// probabilities are then rounded to f16 (quantize_f32_f16 of the lhs, batch_matmul.rs:39).  Ends with a barrier.
// The tree is defined on 256 partial sums (partial v = positions v, v + 256, ... in order) whatever NW is: 16 waves share
// the max / exp / division passes (the long-context softmax kernel), the sums are the 4-wave kernel's bit for bit.
// s_red: NW floats.
// seq_sum: the row sum stays sequential at ANY length (the strict-order device: softmax.rs:43-48 is one scalar loop; the block
// tree beyond 1024 positions is the fast kernels' re-association, pinned in tests/helpers.FAST_TOL)
template <bool F16, int NW = 4>
__device__ __forceinline__ void softmax_row(float* scores, int seq, const unsigned short* __restrict__ exp_tab, float* s_red,
                                            float* s_val_p, bool seq_sum = false) {
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  constexpr int BD = NW * 64;
  float mx = -INFINITY;
  for (int t = tid; t < seq; t += BD) mx = fmaxf(mx, scores[t]);
  mx = wave_max_f32(mx);
  if (lane == 0) s_red[wave] = mx;
  __syncthreads();
  mx = s_red[0];
#pragma unroll
  for (int w = 1; w < NW; w++) mx = fmaxf(mx, s_red[w]);
  __syncthreads();
  float part = 0.0f;
  {
    // the table lookups are independent global gathers: 8 (4) in flight per thread (long rows), summed in t order
    int t = tid;
    for (; t + 7 * BD < seq; t += 8 * BD) {
      float ev[8];
#pragma unroll
      for (int u = 0; u < 8; u++) ev[u] = exp_cached_f(scores[t + u * BD] - mx, exp_tab);
#pragma unroll
      for (int u = 0; u < 8; u++) {
        scores[t + u * BD] = ev[u];
        part += ev[u];
      }
    }
    for (; t + 3 * BD < seq; t += 4 * BD) {
      float ev[4];
#pragma unroll
      for (int u = 0; u < 4; u++) ev[u] = exp_cached_f(scores[t + u * BD] - mx, exp_tab);
#pragma unroll
      for (int u = 0; u < 4; u++) {
        scores[t + u * BD] = ev[u];
        part += ev[u];
      }
    }
    for (; t < seq; t += BD) {
      float ev = exp_cached_f(scores[t] - mx, exp_tab);
      scores[t] = ev;
      part += ev;
    }
  }
  __syncthreads();
  if (seq <= 1024 || seq_sum) {
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
    if constexpr (NW != 4) {  // the 256 partial sums of the 4-wave tree, from the exponentials in LDS
      part = 0.0f;
      if (tid < 256)
        for (int t = tid; t < seq; t += 256) part += scores[t];
    }
    if (NW == 4 || tid < 256) {
      part = wave_sum_f32(part);
      if (lane == 0) s_red[wave] = part;
    }
    __syncthreads();
    if (tid == 0) *s_val_p = (s_red[0] + s_red[1]) + (s_red[2] + s_red[3]);
  }
  __syncthreads();
  const float sum = *s_val_p;
  for (int t = tid; t < seq; t += BD) {
    float pv = scores[t] / sum;
    scores[t] = F16 ? h2f(f2h(pv)) : pv;  // quantize_f32_f16 of the lhs (batch_matmul.rs:39), done once
  }
  __syncthreads();
}

}  // namespace crabml_hip

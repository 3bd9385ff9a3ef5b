// gemv_core.hpp -- device building blocks shared by gemv.hip (one launch per matmul_vec) and fused.hip
// (the fused decode step): exact integer block dots and per-format block load / term evaluation.
#pragma once
#include "devutil.hpp"

namespace crabml_hip {

// ---- exact integer block dots ----------------------------------------------------------------
// Q4_0 block (16 bytes: byte j = elem j (low nibble) | elem j+16 (high nibble)) . 32 int8, minus 8*sum(x)
__device__ __forceinline__ int dot_q4_0(i32x4 q, i32x4 xlo, i32x4 xhi, int xsum) {
  int s = 0;
#pragma unroll
  for (int i = 0; i < 4; i++) {
    int w = q[i];
    s = __builtin_amdgcn_sdot4(w & 0x0F0F0F0F, xlo[i], s, false);
    s = __builtin_amdgcn_sdot4((w >> 4) & 0x0F0F0F0F, xhi[i], s, false);
  }
  return s - 8 * xsum;
}
// unsigned nibbles (Q4_1, Q4_K)
__device__ __forceinline__ int dot_u4(i32x4 q, i32x4 xlo, i32x4 xhi) {
  int s = 0;
#pragma unroll
  for (int i = 0; i < 4; i++) {
    int w = q[i];
    s = __builtin_amdgcn_sdot4(w & 0x0F0F0F0F, xlo[i], s, false);
    s = __builtin_amdgcn_sdot4((w >> 4) & 0x0F0F0F0F, xhi[i], s, false);
  }
  return s;
}
__device__ __forceinline__ int dot_i8x32(i32x4 a0, i32x4 a1, i32x4 b0, i32x4 b1) {
  int s = 0;
#pragma unroll
  for (int i = 0; i < 4; i++) {
    s = __builtin_amdgcn_sdot4(a0[i], b0[i], s, false);
    s = __builtin_amdgcn_sdot4(a1[i], b1[i], s, false);
  }
  return s;
}

// quantized-activation plane views (see common.hpp)
struct ActQ8_0 {
  const i32x4* q;
  const unsigned short* d;
  const int* isum;
};
struct ActQ8_1 {
  const i32x4* q;
  const unsigned short* d;
  const unsigned short* s;
};
struct ActQ8_K {
  const i32x4* q;   // quants in element order (Q5_K, Q6_K, Q2_K, Q3_K, Q8_K weights)
  const float* d;
  const short* bsums;
  const i32x4* qp;  // the same quants class-major inside every 32-element group (common.hpp): what the Q4_K kernels read
};
// the planes of one Q8_K vector (act_layout(Q8_K, n): q | d | bsums | qp)
__host__ __device__ inline ActQ8_K act_q8k_at(const char* planes, size_t off_d, size_t off_aux, size_t off_p) {
  return ActQ8_K{(const i32x4*)planes, (const float*)(planes + off_d), (const short*)(planes + off_aux), (const i32x4*)(planes + off_p)};
}

// ---- the fast step's hop-free norm: 1 / rms applied by the CONSUMER of the quantized row ------------------------------------
// RMSNorm divides the whole row by one number, and the truncating rhs quantizer's levels q = trunc(v / (max|v| / 127))
// (buf_q8_0.rs:119-124) do not depend on a common factor of v.  So the wo / ffn_down launch that produces the residual row x
// quantizes x * w_norm block by block (no in-launch gather of the row's sum of squares), stores the block scale f16(max|x w| /
// 127) in the ordinary d plane together with its chunk's sum of squares of x, and every consuming GEMV multiplies its finished
// row dots by 1 / rms -- a common factor of every block term -- formed from the chunk sums (each wave the same sum in the same
// order; requested when the wave starts, reduced when its dots are done).  Against the exact form: the levels agree up to the
// 126-vs-127 rounding of a block's largest element (which every re-associated GEMV sum of the fast step re-rolls anyway), the
// block scales are f16(max|x w| / 127) / rms where the reference has f16(max|x w| / rms / 127): one f16 rounding either way.
struct RmsTail {
  const float* sums;  // n_sums partial sums of squares of the row, in chunk order
  int n_sums;
  float inv_n, eps;
};
struct RmsReq {
  float v0, v1;
};
// (unconditional loads of clamped indices: a load inside a lane-predicated branch makes the compiler's wait-count pass drain every
// outstanding load at the next divergent join -- in k_qkv that stalled each wave for a memory round trip before its first
// weight request; the lanes past the row's chunks are masked when the values are used)
__device__ __forceinline__ RmsReq rms_request(const RmsTail& t, int lane) {
  const int last = t.n_sums - 1;
  return RmsReq{t.sums[lane < last ? lane : last], t.sums[lane + 64 < last ? lane + 64 : last]};
}
// The tail of a consuming wave: kept short (one DPP tree, the hardware's reciprocal square root) because every wave of a launch
// reaches it at about the same time -- nothing hides it.  A fixed lane-strided order + v_rsq_f32: every wave of every consumer
// forms the same bits (the scale is a common factor; its last-ulp value is not the reference's -- nor is anything else here).
__device__ __forceinline__ float rms_finish(const RmsTail& t, const RmsReq& r, int lane) {
  float part = (lane < t.n_sums ? r.v0 : 0.0f) + (lane + 64 < t.n_sums ? r.v1 : 0.0f);
  for (int base = 128; base < t.n_sums; base += 64)  // rows past 4096 elements
    part += base + lane < t.n_sums ? t.sums[base + lane] : 0.0f;
  return __builtin_amdgcn_rsqf(wave_sum_f32(part) * t.inv_n + t.eps);
}

// ---- per-format access for the 32-element formats whose rhs is Q8_0 --------------------------------------
// A lane processes one 16-byte UNIT of quants per step, so that a wave's weight load is always one aligned 1 KiB
// request: a whole Q4_0 block (32 nibbles), or HALF a Q8_0 block (16 int8; the two lanes of a block add their
// integer partials with one DPP swap, so the block's `sumi` stays exact and only the even lane contributes the
// scaled term).  A row has nu = nb * UNITS units; unit u of row r is quant word r * nu + u and uses the scale of
// block u / UNITS.
struct XUnit {  // the activation side of one unit
  i32x4 x0, x1;
  float dx;
  int xs;  // Q4_0: sum of the block's quants; Q4_1: the Q8_1 block's raw f16 pair d | s << 16
};
template <int FMT>
struct BlockFmt;

template <>
struct BlockFmt<CRABML_HIP_Q4_0> {
  static constexpr int UNITS = 1;
  struct Blk {
    i32x4 q;
    unsigned short d;
  };
  static __device__ __forceinline__ Blk load(const i32x4* wq, const unsigned short* wd, size_t row, int nb, int u) {
    Blk b;
    b.q = __builtin_nontemporal_load(wq + row * nb + u);
    b.d = __builtin_nontemporal_load(wd + row * nb + u);
    return b;
  }
  static __device__ __forceinline__ XUnit loadx(const ActQ8_0& act, int u) {
    return XUnit{act.q[2 * u], act.q[2 * u + 1], h2f(act.d[u]), act.isum[u]};
  }
  // buf_q4_0.rs:249: sumi as f32 * d_w * d_x
  static __device__ __forceinline__ float term(const Blk& b, const XUnit& x) {
    return ((float)dot_q4_0(b.q, x.x0, x.x1, x.xs) * h2f(b.d)) * x.dx;
  }
};

template <>
struct BlockFmt<CRABML_HIP_Q8_0> {
  static constexpr int UNITS = 2;
  struct Blk {
    i32x4 q;
    unsigned short d;
  };
  static __device__ __forceinline__ Blk load(const i32x4* wq, const unsigned short* wd, size_t row, int nb, int u) {
    Blk b;
    b.q = __builtin_nontemporal_load(wq + row * (2 * (size_t)nb) + u);
    b.d = __builtin_nontemporal_load(wd + row * nb + (u >> 1));
    return b;
  }
  static __device__ __forceinline__ XUnit loadx(const ActQ8_0& act, int u) {
    return XUnit{act.q[u], i32x4{0, 0, 0, 0}, h2f(act.d[u >> 1]), 0};
  }
  // buf_q8_0.rs:282; lanes 2j / 2j+1 hold the two halves of a block (both active: nu is even)
  static __device__ __forceinline__ float term(const Blk& b, const XUnit& x) {
    int s = 0;
#pragma unroll
    for (int i = 0; i < 4; i++) s = __builtin_amdgcn_sdot4(b.q[i], x.x0[i], s, false);
    s += dpp_i<0xB1>(s);  // quad_perm [1,0,3,2]: the partner's half
    const float t = ((float)s * h2f(b.d)) * x.dx;
    return (threadIdx.x & 1) ? 0.0f : t;
  }
};

template <>
struct BlockFmt<CRABML_HIP_Q4_1> {  // planes qs[n][16] | (d, m)[n] f16 pairs; rhs Q8_1 (q | d | s)
  static constexpr int UNITS = 1;
  struct Blk {
    i32x4 q;
    unsigned dm;
  };
  static __device__ __forceinline__ Blk load(const i32x4* wq, const unsigned short* wd, size_t row, int nb, int u) {
    Blk b;
    b.q = __builtin_nontemporal_load(wq + row * nb + u);
    b.dm = __builtin_nontemporal_load((const unsigned*)wd + row * nb + u);
    return b;
  }
  static __device__ __forceinline__ XUnit loadx(const ActQ8_1& act, int u) {
    return XUnit{act.q[2 * u], act.q[2 * u + 1], 0.0f, (int)((unsigned)act.d[u] | ((unsigned)act.s[u] << 16))};
  }
  // buf_q4_1.rs:276: (d_w * d_x) and (m * s) are f16 products rounded to f16 by the half crate
  static __device__ __forceinline__ float term(const Blk& b, const XUnit& x) {
    const int si = dot_u4(b.q, x.x0, x.x1);
    const unsigned short dwh = (unsigned short)(b.dm & 0xffffu), mwh = (unsigned short)(b.dm >> 16);
    const unsigned short dxh = (unsigned short)((unsigned)x.xs & 0xffffu), sxh = (unsigned short)((unsigned)x.xs >> 16);
    return h2f(h_mul(dwh, dxh)) * (float)si + h2f(h_mul(mwh, sxh));
  }
};

// Per-lane partial sums of R rows of one weight matrix against one quantized activation vector: lane l owns
// units l, l+64, ...  The R unit loads of a step are issued before any is consumed.
template <int FMT, int R, class ACT>
__device__ __forceinline__ void rows_partial(const i32x4* __restrict__ wq, const unsigned short* __restrict__ wd,
                                             const ACT& act, int row0, int m, int nb, int lane, float acc[R]) {
  using F = BlockFmt<FMT>;
#pragma unroll
  for (int r = 0; r < R; r++) acc[r] = 0.f;
  const int nu = nb * F::UNITS;
  for (int u = lane; u < nu; u += 64) {
    typename F::Blk blk[R];
#pragma unroll
    for (int r = 0; r < R; r++) {
      int row = row0 + r < m ? row0 + r : m - 1;
      blk[r] = F::load(wq, wd, (size_t)row, nb, u);
    }
    const XUnit x = F::loadx(act, u);
#pragma unroll
    for (int r = 0; r < R; r++) acc[r] += F::term(blk[r], x);
  }
}

// rows_partial with TWO 64-unit steps requested per round (rows of a multiple of 128 units): for launches of few, short waves -- a
// tensor-parallel rank's q|k|v: 640 waves of four steps -- halving a wave's dependent round trips shortens the launch; with many
// resident waves it only delays each wave's first use (profiles/r06_small_stage_ab.md).  Same terms, same per-lane order.
template <int FMT, int R, class ACT>
__device__ __forceinline__ void rows_partial_2step(const i32x4* __restrict__ wq, const unsigned short* __restrict__ wd, const ACT& act,
                                                   int row0, int m, int nb, int lane, float acc[R]) {
  using F = BlockFmt<FMT>;
#pragma unroll
  for (int r = 0; r < R; r++) acc[r] = 0.f;
  const int nu = nb * F::UNITS;  // (a multiple of 128: the caller checked)
  for (int u = lane; u < nu; u += 128) {
    typename F::Blk b0[R], b1[R];
#pragma unroll
    for (int r = 0; r < R; r++) {
      const int row = row0 + r < m ? row0 + r : m - 1;
      b0[r] = F::load(wq, wd, (size_t)row, nb, u);
      b1[r] = F::load(wq, wd, (size_t)row, nb, u + 64);
    }
    const XUnit x0 = F::loadx(act, u), x1 = F::loadx(act, u + 64);
#pragma unroll
    for (int r = 0; r < R; r++) {
      acc[r] += F::term(b0[r], x0);
      acc[r] += F::term(b1[r], x1);
    }
  }
}

// rows_partial for a wave that also has to form 1 / rms from the chunk sums it requested when it started (RmsTail): a uniform trip
// count (lanes past the row's units redo the last one and add nothing) so that the whole wave can run the reduction INSIDE the
// first step -- behind that step's weight requests, while they are in flight.  After its last step a wave is on the launch's
// critical path; in the first step it is waiting for memory anyway.
template <int FMT, int R, class ACT>
__device__ __forceinline__ float rows_partial_rms(const i32x4* __restrict__ wq, const unsigned short* __restrict__ wd, const ACT& act,
                                                  int row0, int m, int nb, int lane, float acc[R], const RmsTail& rt, RmsReq rq) {
  using F = BlockFmt<FMT>;
#pragma unroll
  for (int r = 0; r < R; r++) acc[r] = 0.f;
  const int nu = nb * F::UNITS;
  float inv_rms = 1.0f;
  for (int u0 = 0; u0 < nu; u0 += 64) {
    const int u = u0 + lane;
    const bool live = u < nu;  // (Q8_0: nu is even, the two lanes of a block are live or dead together)
    const int uu = live ? u : nu - 1;
    typename F::Blk blk[R];
#pragma unroll
    for (int r = 0; r < R; r++) {
      int row = row0 + r < m ? row0 + r : m - 1;
      blk[r] = F::load(wq, wd, (size_t)row, nb, uu);
    }
    const XUnit x = F::loadx(act, uu);
    if (u0 == 0) {
      // (pinned behind this step's requests: the empty asm keeps the reduction from being hoisted out of the loop, the scheduling
      // barrier from being moved above the loads)
      __builtin_amdgcn_sched_barrier(0);
      asm volatile("" : "+v"(rq.v0), "+v"(rq.v1));
      inv_rms = rms_finish(rt, rq, lane);
    }
#pragma unroll
    for (int r = 0; r < R; r++) {
      const float t = F::term(blk[r], x);
      acc[r] += live ? t : 0.0f;
    }
  }
  return inv_rms;
}

// The same for rows of exactly 128 units (k = 4096 in Q4_0: q / k / v of the 8B shape) with BOTH 64-unit steps requested up front --
// one round of requests per wave instead of two dependent ones.  Same terms, same per-lane order (step 0 then step 1): bit-identical
// to rows_partial_rms (k_qkv's default for such rows; profiles/r06_small_stage_ab.md).
template <int FMT, int R, class ACT>
__device__ __forceinline__ float rows_partial_rms_128(const i32x4* __restrict__ wq, const unsigned short* __restrict__ wd, const ACT& act,
                                                      int row0, int m, int nb, int lane, float acc[R], const RmsTail& rt, RmsReq rq) {
  using F = BlockFmt<FMT>;
  typename F::Blk b0[R], b1[R];
#pragma unroll
  for (int r = 0; r < R; r++) {
    const int row = row0 + r < m ? row0 + r : m - 1;
    b0[r] = F::load(wq, wd, (size_t)row, nb, lane);
    b1[r] = F::load(wq, wd, (size_t)row, nb, lane + 64);
  }
  const XUnit x0 = F::loadx(act, lane), x1 = F::loadx(act, lane + 64);
  __builtin_amdgcn_sched_barrier(0);
  asm volatile("" : "+v"(rq.v0), "+v"(rq.v1));
  const float inv_rms = rms_finish(rt, rq, lane);
#pragma unroll
  for (int r = 0; r < R; r++) {
    acc[r] = 0.f;
    acc[r] += F::term(b0[r], x0);
    acc[r] += F::term(b1[r], x1);
  }
  return inv_rms;
}

// ---- strict order at streaming speed: the block terms of R rows into a term table, then one lane per row adds them in block order ----
// (CRABML_HIP_FLAG_STRICT_ORDER.)  The reference's scalar dot of these formats is `sumf = 0; for block: sumf += term(block)` with one
// f32 term per block (buf_q4_0.rs:240-253, buf_q8_0.rs:275-286, buf_q4_1.rs:266-280): the terms are evaluated exactly as the fast
// kernels do (same loads, same integers, the reference's expression per block), parked in LDS -- T[r * stride + block] -- and added by
// ordered_sum.  Bit-identical to the one-thread-per-row loop (k_gemv_strict) and to the oracle.
template <int FMT, int R, class ACT, bool UPFRONT = false>
__device__ __forceinline__ void rows_terms(const i32x4* __restrict__ wq, const unsigned short* __restrict__ wd, const ACT& act, int row0,
                                           int m, int nb, int lane, float* __restrict__ T, int stride) {
  using F = BlockFmt<FMT>;
  const int nu = nb * F::UNITS;
  if (UPFRONT && nu == 128) {  // (uniform) rows of exactly two 64-unit steps: one round of requests (k_qkv_ord; as rows_partial_rms_128)
    typename F::Blk b0[R], b1[R];
#pragma unroll
    for (int r = 0; r < R; r++) {
      const int row = row0 + r < m ? row0 + r : m - 1;
      b0[r] = F::load(wq, wd, (size_t)row, nb, lane);
      b1[r] = F::load(wq, wd, (size_t)row, nb, lane + 64);
    }
    const XUnit x0 = F::loadx(act, lane), x1 = F::loadx(act, lane + 64);
#pragma unroll
    for (int r = 0; r < R; r++) {
      const float t0 = F::term(b0[r], x0), t1 = F::term(b1[r], x1);
      if (F::UNITS == 1 || (lane & 1) == 0) {
        T[r * stride + lane / F::UNITS] = t0;
        T[r * stride + (lane + 64) / F::UNITS] = t1;
      }
    }
    return;
  }
  for (int u0 = 0; u0 < nu; u0 += 64) {
    const int u = u0 + lane;
    const bool live = u < nu;  // (Q8_0: nu is even, the two lanes of a block are live or dead together)
    const int uu = live ? u : nu - 1;
    typename F::Blk blk[R];
#pragma unroll
    for (int r = 0; r < R; r++) {
      const int row = row0 + r < m ? row0 + r : m - 1;
      blk[r] = F::load(wq, wd, (size_t)row, nb, uu);
    }
    const XUnit x = F::loadx(act, uu);
#pragma unroll
    for (int r = 0; r < R; r++) {
      const float t = F::term(blk[r], x);
      if (live && (F::UNITS == 1 || (lane & 1) == 0)) T[r * stride + uu / F::UNITS] = t;
    }
  }
}
// t: 16-byte aligned
__device__ __forceinline__ float ordered_sum(const float* __restrict__ t, int nterms) {
  float sumf = 0.0f;
  int i = 0;
  for (; i + 4 <= nterms; i += 4) {
    const f32x4 v = *(const f32x4*)(t + i);
    sumf += v[0];
    sumf += v[1];
    sumf += v[2];
    sumf += v[3];
  }
  for (; i < nterms; i++) sumf += t[i];
  return sumf;
}

// Q4_K rows (planes qs[n][128] | hdr[n][16]) against a Q8_K activation vector: lane = one 16-byte qs piece j of a
// super-block (8 lanes per super-block: a wave's load is one aligned 1 KiB request); piece j belongs to the
// 64-element pair p = j / 2 and carries the low nibbles of sub-block 2p and the high nibbles of sub-block 2p + 1
// (buf_q4_k.rs:212-217) for the element classes 4 (j & 1) .. +4 -- dword i of the piece = the four elements e of the
// 32-group with e % 8 == 4 (j & 1) + i (class-major planes, common.hpp; the activation plane `qp` is laid out the same way).  The 16-byte header is shared by the 8 lanes; its
// (scale, min) fields were re-packed pair-major at upload (common.hpp), so the lane's four 6-bit values are one
// funnel shift and four bit-field extracts.
// HDR_DPP: each lane loads ONE dword of the 16-byte header (lane & 3 selects it) and the quad exchanges the four
// dwords with DPP -- 4 instead of 16 header bytes per lane through the vector-memory path (measured: down 12.9 ->
// 11.5 us, classifier 53.7 -> 49.3 us); with the activation planes in LDS that path is no longer the bottleneck
// and the plain 16-byte load is faster (gate/up 15.9 vs 17.1 us), hence the switch.
template <bool HDR_DPP>
struct Q4KPiece {  // one lane's 16-byte qs piece + its super-block header (whole, or the quad's dword of it)
  i32x4 qv, hdr;
  unsigned hw;
};
template <bool HDR_DPP>
__device__ __forceinline__ Q4KPiece<HDR_DPP> q4k_load(const i32x4* __restrict__ wq, const i32x4* __restrict__ wh, size_t row, int nsb,
                                                      int c, int lane) {
  Q4KPiece<HDR_DPP> w;
  const int sb = c >> 3;
  w.qv = __builtin_nontemporal_load(wq + row * (size_t)(nsb * 8) + c);
  if constexpr (HDR_DPP)
    w.hw = __builtin_nontemporal_load((const unsigned*)wh + (row * nsb + sb) * 4 + (lane & 3));
  else
    w.hdr = __builtin_nontemporal_load(wh + row * nsb + sb);
  return w;
}
struct Q4KX {  // the activation side of piece c
  i32x4 xl, xh;
  float d8;
  int bs_lo, bs_hi;
};
template <bool CM = true>  // CM: the class-major plane (Q4_K weights); else element order (Q5_K, whose planes keep the file's order)
__device__ __forceinline__ Q4KX q4k_loadx(const ActQ8_K& act, int c) {
  const int sb = c >> 3, j = c & 7, p = j >> 1, h = j & 1;
  Q4KX x;
  const i32x4* xq = (CM ? act.qp : act.q) + (size_t)sb * 16 + p * 4 + h;  // class-major: dwords 4 h .. 4 h + 3 of each 32-group = classes 4 h ..
  x.xl = xq[0];
  x.xh = xq[2];
  x.d8 = act.d[sb];
  // (the minimum term only needs every bsums entry taken once per super-block: the piece keeps the two it always took)
  const short* bs = act.bsums + sb * 16 + p * 4 + h;
  x.bs_lo = (int)bs[0];
  x.bs_hi = (int)bs[2];
  return x;
}
// The exact integer part of piece c (buf_q4_k.rs:212-263): isum = sc_lo * sum(q4 q8 | low nibbles) + sc_hi * sum(q4 q8 | high
// nibbles), msum = m_lo * bsum_lo + m_hi * bsum_hi; h0 = the header's first dword (d | dmin << 16).  Every Q4_K GEMV kernel
// (k_gemv_q4_k, k_qkv, k_gemv_res_nq, k_gateup_k_lds) gets its integers from here; crabml_hip_debug_superblock_ints dumps them.
template <bool HDR_DPP>
__device__ __forceinline__ void q4k_ints(const Q4KPiece<HDR_DPP>& w, const Q4KX& x, int c, int& isum, int& msum, unsigned& h0) {
  const int p = (c & 7) >> 1;
  unsigned h1, h2, h3;
  if constexpr (HDR_DPP) {  // quad_perm broadcasts of dword 0..3
    h0 = (unsigned)dpp_i<0x00>((int)w.hw);
    h1 = (unsigned)dpp_i<0x55>((int)w.hw);
    h2 = (unsigned)dpp_i<0xAA>((int)w.hw);
    h3 = (unsigned)dpp_i<0xFF>((int)w.hw);
  } else {
    h0 = (unsigned)w.hdr[0];
    h1 = (unsigned)w.hdr[1];
    h2 = (unsigned)w.hdr[2];
    h3 = (unsigned)w.hdr[3];
  }
  const unsigned f = q4k_pair_field(h1, h2, h3, p);
  const int sc_lo = (int)(f & 63u), sc_hi = (int)((f >> 6) & 63u);
  const int m_lo = (int)((f >> 12) & 63u), m_hi = (int)(f >> 18);
  int lo = 0, hi = 0;
#pragma unroll
  for (int i = 0; i < 4; i++) {
    lo = __builtin_amdgcn_sdot4(w.qv[i] & 0x0F0F0F0F, x.xl[i], lo, false);
    hi = __builtin_amdgcn_sdot4((w.qv[i] >> 4) & 0x0F0F0F0F, x.xh[i], hi, false);
  }
  // (v_mul_i32_i24 / v_mad_i32_i24: every factor is below 2^23 -- |lo|, |hi| <= 16 * 15 * 128, the 6-bit fields, |bsum| <= 16 * 128 --
  // where a plain 32-bit multiply is a quarter-rate instruction)
  isum = __mul24(sc_lo, lo) + __mul24(sc_hi, hi);          // exact (the reference's aux32 lanes hold integers < 2^24)
  msum = __mul24(m_lo, x.bs_lo) + __mul24(m_hi, x.bs_hi);  // i32: the intended math of buf_q4_k.rs:238-241
}
template <bool HDR_DPP>
__device__ __forceinline__ float q4k_term(const Q4KPiece<HDR_DPP>& w, const Q4KX& x, int c, int* dbg = nullptr) {
  int isum, msum;
  unsigned h0;
  q4k_ints<HDR_DPP>(w, x, c, isum, msum, h0);
  if (dbg != nullptr) {  // parity hook (DBG instantiations only): this piece's integers, as the float part consumes them
    dbg[0] = isum;
    dbg[1] = msum;
  }
  const float dd = h2f((unsigned short)(h0 & 0xffff)) * x.d8;
  const float dmin = h2f((unsigned short)(h0 >> 16)) * x.d8;
  return dd * (float)isum - dmin * (float)msum;
}
// ---- Q4_K in the reference's order (strict-order device) ---------------------------------------------------------------------
// buf_q4_k.rs:192-277 keeps EIGHT f32 lanes per row: inside a super-block `aux32[l] += scale * (q8 * q4)` for the elements e of every
// 32-group with e % 8 == l -- sums of integers below 2^24, exact in f32 in any order --, then per super-block `sums[l] += d * aux32[l]`
// and `sumf -= dmin * sumi`, and at the end `sumf += sums[0..8)` in order.  A super-block therefore contributes nine f32 terms per row:
// d * A[l] with A[l] the exact integer lane sum, and dmin * sumi.  With the planes class-major (common.hpp) dword i of piece (p, h) IS
// class 4 h + i of pair p: one v_dot4 per nibble half and class, A[4 h + i] = sum over the four pairs (the four lanes of the
// super-block with the same h: lane ^ 2 and lane ^ 4 over DPP).  The lane with p = 0 parks its four terms in the super-block's
// 12-float record t (floats 0..7 = d * A[l], 8 = dmin * sumi); q4k_ordered_sum then runs the nine chains over the records in order.
// All 64 lanes converged; `live` = the lane's piece exists (dead lanes contribute zeros and store nothing).
template <bool HDR_DPP>
__device__ __forceinline__ void q4k_class_terms(const Q4KPiece<HDR_DPP>& w, const Q4KX& x, int c, bool live, float* __restrict__ t) {
  const int j = c & 7, p = j >> 1, h = j & 1;
  unsigned h0, h1, h2, h3;
  if constexpr (HDR_DPP) {
    h0 = (unsigned)dpp_i<0x00>((int)w.hw);
    h1 = (unsigned)dpp_i<0x55>((int)w.hw);
    h2 = (unsigned)dpp_i<0xAA>((int)w.hw);
    h3 = (unsigned)dpp_i<0xFF>((int)w.hw);
  } else {
    h0 = (unsigned)w.hdr[0];
    h1 = (unsigned)w.hdr[1];
    h2 = (unsigned)w.hdr[2];
    h3 = (unsigned)w.hdr[3];
  }
  const unsigned f = q4k_pair_field(h1, h2, h3, p);
  const int sc_lo = (int)(f & 63u), sc_hi = (int)((f >> 6) & 63u);
  const int m_lo = (int)((f >> 12) & 63u), m_hi = (int)(f >> 18);
  int A[4];
#pragma unroll
  for (int i = 0; i < 4; i++) {
    const int lo = __builtin_amdgcn_sdot4(w.qv[i] & 0x0F0F0F0F, x.xl[i], 0, false);
    const int hi = __builtin_amdgcn_sdot4((w.qv[i] >> 4) & 0x0F0F0F0F, x.xh[i], 0, false);
    int a = live ? __mul24(sc_lo, lo) + __mul24(sc_hi, hi) : 0;  // (24-bit factors: full-rate multiplies)
    a += dpp_i<0x4E>(a);   // lane ^ 2 (quad_perm [2,3,0,1])
    a += dpp_i<0x104>(a);  // row_shl:4 -- lane j takes lane j + 4 of its row: the p = 0 lanes (j < 2 of every 8) hold the four pairs' sum
    A[i] = a;
  }
  int ms = live ? __mul24(m_lo, x.bs_lo) + __mul24(m_hi, x.bs_hi) : 0;  // i32: the intended math of buf_q4_k.rs:238-241
  ms += dpp_i<0xB1>(ms);
  ms += dpp_i<0x4E>(ms);
  ms += dpp_i<0x141>(ms);
  if (live && p == 0) {
    const float d = h2f((unsigned short)(h0 & 0xffff)) * x.d8;
    *(f32x4*)(t + 4 * h) = f32x4{d * (float)A[0], d * (float)A[1], d * (float)A[2], d * (float)A[3]};
    if (h == 0) t[8] = (h2f((unsigned short)(h0 >> 16)) * x.d8) * (float)ms;
  }
}
// floats per row of a record table: 12 per super-block + 4 of padding, so that the chain lanes (one per row) read their 16-byte pieces
// four LDS banks apart (a stride of 12 nsb floats is a multiple of 64 banks for nsb = 16: every lane on the same banks)
__host__ __device__ inline int q4k_rec_stride(int nsb) { return nsb * 12 + 4; }
// the nine chains of one row over its nsb records (12 floats each, 16-byte aligned), in super-block order: buf_q4_k.rs:263-276
__device__ __forceinline__ float q4k_ordered_sum(const float* __restrict__ t, int nsb) {
  float sums[8], sumf = 0.0f;
#pragma unroll
  for (int l = 0; l < 8; l++) sums[l] = 0.0f;
  // four records' reads in flight ahead of their adds (the nine chains are independent of each other: one dependent add per record)
  int sb = 0;
  for (; sb + 4 <= nsb; sb += 4) {
    f32x4 a[4], b[4];
    float m[4];
#pragma unroll
    for (int u = 0; u < 4; u++) {
      a[u] = *(const f32x4*)(t + (sb + u) * 12);
      b[u] = *(const f32x4*)(t + (sb + u) * 12 + 4);
      m[u] = t[(sb + u) * 12 + 8];
    }
#pragma unroll
    for (int u = 0; u < 4; u++) {
      sums[0] += a[u][0];
      sums[1] += a[u][1];
      sums[2] += a[u][2];
      sums[3] += a[u][3];
      sums[4] += b[u][0];
      sums[5] += b[u][1];
      sums[6] += b[u][2];
      sums[7] += b[u][3];
      sumf -= m[u];
    }
  }
  for (; sb < nsb; sb++) {
    const f32x4 a = *(const f32x4*)(t + sb * 12), b = *(const f32x4*)(t + sb * 12 + 4);
    sums[0] += a[0];
    sums[1] += a[1];
    sums[2] += a[2];
    sums[3] += a[3];
    sums[4] += b[0];
    sums[5] += b[1];
    sums[6] += b[2];
    sums[7] += b[3];
    sumf -= t[sb * 12 + 8];
  }
#pragma unroll
  for (int l = 0; l < 8; l++) sumf += sums[l];
  return sumf;
}
// the records of R rows of a Q4_K matrix: T[r * stride + sb * 12 ..]; lane = one 16-byte piece, as rows_partial_q4k
template <int R, bool HDR_DPP>
__device__ __forceinline__ void rows_terms_q4k(const i32x4* __restrict__ wq, const i32x4* __restrict__ wh, const ActQ8_K& act, int row0, int m,
                                               int nsb, int lane, float* __restrict__ T, int stride, int cfirst = 0) {
  const int np = nsb * 8;
  for (int c0 = cfirst; c0 < np; c0 += 64) {  // cfirst: pieces below it were taken by the caller (a multiple of 64)
    const int c = c0 + lane;
    const bool live = c < np;  // (a super-block's eight lanes are live or dead together)
    const int cc = live ? c : np - 1;
    Q4KPiece<HDR_DPP> w[R];
#pragma unroll
    for (int r = 0; r < R; r++) w[r] = q4k_load<HDR_DPP>(wq, wh, (size_t)(row0 + r < m ? row0 + r : m - 1), nsb, cc, lane);
    const Q4KX x = q4k_loadx(act, cc);
#pragma unroll
    for (int r = 0; r < R; r++) q4k_class_terms<HDR_DPP>(w[r], x, cc, live, T + (size_t)r * stride + (cc >> 3) * 12);
  }
}

// The same records for R rows of a Q6_K matrix standing in a Q4_K layer (attn_v / ffn_down of the *_K_M mixes) on the strict-order
// device (buf_q6_k.rs:183-234: eight f32 lanes, element e feeds lane e % 8 -- exact integers scale * q8 * (q6 - 32) summed per
// super-block, `sums[l] += aux32[l] * d` in super-block order, the lanes added in order at the end; k_gemv_exact_q6k's arithmetic).
// A record is Q4_K's 12 floats with float 8 = +0.0 (Q6_K has no minimum term: q4k_ordered_sum's `sumf -= 0.0` changes no bit), so the
// ordered launches keep ONE table layout and ONE chain.  planes ql | qh | scales | d; act: Q8_K planes in ELEMENT order (act.q);
// lane = one 16-byte ql piece (8 per super-block); the levels are made signed bytes (q6 - 32), the v_dot4 sums split by byte position.
template <int R>
__device__ __forceinline__ void rows_terms_q6k(const char* __restrict__ w, size_t off_qh, const ActQ8_K& act, int row0, int m, int nsb, int lane,
                                               float* __restrict__ T, int stride) {
  const size_t n = off_qh / 128;
  const i32x4* wql = (const i32x4*)w;
  const i32x4* wqh = (const i32x4*)(w + off_qh);
  const i32x4* wsc = (const i32x4*)(w + off_qh + n * 64);
  const unsigned short* wd = (const unsigned short*)(w + off_qh + n * 80);
  const int np = nsb * 8;
  for (int c0 = 0; c0 < np; c0 += 64) {
    const int c = c0 + lane;
    const bool live = c < np;
    const int cc = live ? c : np - 1;
    const int sb = cc >> 3, h = (cc >> 2) & 1, a = (cc >> 1) & 1, p = cc & 1;
    const int gi = 8 * h + p + 2 * a;  // the low nibbles' 16-element scale group; the high nibbles' is gi + 4
    const i32x4* xq = act.q + (size_t)sb * 16 + gi;
    const i32x4 xl = xq[0], xh = xq[4];
    const float d8 = act.d[sb];
#pragma unroll
    for (int r = 0; r < R; r++) {
      const size_t blk = (size_t)(row0 + r < m ? row0 + r : m - 1) * nsb + sb;
      const i32x4 qv = __builtin_nontemporal_load(wql + blk * 8 + (cc & 7));
      const i32x4 hv = __builtin_nontemporal_load(wqh + blk * 4 + 2 * h + p);
      const i32x4 sc4 = __builtin_nontemporal_load(wsc + blk);
      const int sc_lo = (int)(signed char)(((unsigned)sc4[gi >> 2] >> (8 * (gi & 3))) & 0xFFu);
      const int sc_hi = (int)(signed char)(((unsigned)sc4[(gi + 4) >> 2] >> (8 * (gi & 3))) & 0xFFu);
      int lo[8], hi[8];
#pragma unroll
      for (int l = 0; l < 8; l++) lo[l] = hi[l] = 0;
#pragma unroll
      for (int i = 0; i < 4; i++) {
        const unsigned q = (unsigned)qv[i], hb = (unsigned)hv[i] >> (2 * a);
        const unsigned l6 = (q & 0x0F0F0F0Fu) | ((hb & 0x03030303u) << 4);
        const unsigned h6 = ((q >> 4) & 0x0F0F0F0Fu) | (((hb >> 4) & 0x03030303u) << 4);
        // 0 .. 63 -> signed bytes v - 32 without carries between bytes
        const int ls = (int)((((l6 | 0x80808080u) - 0x20202020u)) ^ 0x80808080u);
        const int hs = (int)((((h6 | 0x80808080u) - 0x20202020u)) ^ 0x80808080u);
#pragma unroll
        for (int k = 0; k < 4; k++) {
          lo[4 * (i & 1) + k] = __builtin_amdgcn_sdot4(ls, (int)((unsigned)xl[i] & (0xFFu << (8 * k))), lo[4 * (i & 1) + k], false);
          hi[4 * (i & 1) + k] = __builtin_amdgcn_sdot4(hs, (int)((unsigned)xh[i] & (0xFFu << (8 * k))), hi[4 * (i & 1) + k], false);
        }
      }
      int A[8];
#pragma unroll
      for (int l = 0; l < 8; l++) {
        A[l] = live ? sc_lo * lo[l] + sc_hi * hi[l] : 0;
        A[l] += dpp_i<0xB1>(A[l]);
        A[l] += dpp_i<0x4E>(A[l]);
        A[l] += dpp_i<0x141>(A[l]);
      }
      if (live && (lane & 7) == 0) {
        const float d = h2f(wd[blk]) * d8;
        float* t = T + (size_t)r * stride + sb * 12;
        *(f32x4*)t = f32x4{(float)A[0] * d, (float)A[1] * d, (float)A[2] * d, (float)A[3] * d};
        *(f32x4*)(t + 4) = f32x4{(float)A[4] * d, (float)A[5] * d, (float)A[6] * d, (float)A[7] * d};
        t[8] = 0.0f;
      }
    }
  }
}

template <int R, bool HDR_DPP = true, bool DBG = false>
__device__ __forceinline__ void rows_partial_q4k(const i32x4* __restrict__ wq, const i32x4* __restrict__ wh, const ActQ8_K& act,
                                                 int row0, int m, int nsb, int lane, float acc[R], int c0 = 0, int* dbg = nullptr) {
  if (c0 == 0) {
#pragma unroll
    for (int r = 0; r < R; r++) acc[r] = 0.f;
  }
  const int nchunks = nsb * 8;
  for (int c = c0 + lane; c < nchunks; c += 64) {  // c0: pieces below it were taken by the caller (a multiple of 64)
    Q4KPiece<HDR_DPP> w[R];
#pragma unroll
    for (int r = 0; r < R; r++) {
      int row = row0 + r < m ? row0 + r : m - 1;
      w[r] = q4k_load<HDR_DPP>(wq, wh, (size_t)row, nsb, c, lane);
    }
    const Q4KX x = q4k_loadx(act, c);
#pragma unroll
    for (int r = 0; r < R; r++) {
      if constexpr (DBG)
        acc[r] += q4k_term<HDR_DPP>(w[r], x, c, dbg + ((size_t)r * nchunks + c) * 2);
      else
        acc[r] += q4k_term<HDR_DPP>(w[r], x, c);
    }
  }
}

// Q5_K rows (planes qs[n][128] | qh[n][32] | hdr[n][16], the reference's own block contents: buf_q5_k.rs:13-21) against a Q8_K
// activation vector: Q4_K's mapping (lane = one 16-byte qs piece j of a super-block: pair p = j / 2, positions 16 (j & 1) .. +16)
// plus the piece's 16 bytes of qh, whose bits 2p / 2p + 1 are the fifth bit of the low / high nibbles (buf_q5_k.rs:246-262).
// Levels 0 .. 31 fit the signed bytes of v_dot4; scales, minimums and the float part are Q4_K's.  Not a tuned path (three
// 16-byte loads per lane and piece): Q5_K runs as per-op segments.
template <int R>
__device__ __forceinline__ void rows_partial_q5k(const char* __restrict__ w, size_t off_qh, const ActQ8_K& act, int row0, int m,
                                                 int nsb, int lane, float acc[R]) {
  const size_t n = off_qh / 128;  // blocks in the tensor
  const i32x4* wq = (const i32x4*)w;
  const i32x4* wqh = (const i32x4*)(w + off_qh);
  const i32x4* wh = (const i32x4*)(w + off_qh + n * 32);
#pragma unroll
  for (int r = 0; r < R; r++) acc[r] = 0.f;
  const int nchunks = nsb * 8;
  for (int c = lane; c < nchunks; c += 64) {
    const int sb = c >> 3, j = c & 7, p = j >> 1, h = j & 1;
    const Q4KX x = q4k_loadx<false>(act, c);
#pragma unroll
    for (int r = 0; r < R; r++) {
      const size_t blk = (size_t)(row0 + r < m ? row0 + r : m - 1) * nsb + sb;
      const i32x4 qv = __builtin_nontemporal_load(wq + blk * 8 + j);
      const i32x4 hv = __builtin_nontemporal_load(wqh + blk * 2 + h);
      const i32x4 hd = __builtin_nontemporal_load(wh + blk);
      const unsigned f = q4k_pair_field((unsigned)hd[1], (unsigned)hd[2], (unsigned)hd[3], p);
      const int sc_lo = (int)(f & 63u), sc_hi = (int)((f >> 6) & 63u);
      const int m_lo = (int)((f >> 12) & 63u), m_hi = (int)(f >> 18);
      int lo = 0, hi = 0;
#pragma unroll
      for (int i = 0; i < 4; i++) {
        const unsigned q = (unsigned)qv[i], hb = (unsigned)hv[i] >> (2 * p);
        const unsigned l5 = (q & 0x0F0F0F0Fu) | ((hb & 0x01010101u) << 4);
        const unsigned h5 = ((q >> 4) & 0x0F0F0F0Fu) | (((hb >> 1) & 0x01010101u) << 4);
        lo = __builtin_amdgcn_sdot4((int)l5, x.xl[i], lo, false);
        hi = __builtin_amdgcn_sdot4((int)h5, x.xh[i], hi, false);
      }
      const int isum = sc_lo * lo + sc_hi * hi;
      const int msum = m_lo * x.bs_lo + m_hi * x.bs_hi;
      const unsigned h0 = (unsigned)hd[0];
      const float dd = h2f((unsigned short)(h0 & 0xffff)) * x.d8;
      const float dmin = h2f((unsigned short)(h0 >> 16)) * x.d8;
      acc[r] += dd * (float)isum - dmin * (float)msum;
    }
  }
}

// Q6_K rows (planes ql[n][128] | qh[n][64] | scales[n][16] | d[n] f16; common.hpp) against a Q8_K activation vector:
// lane = one 16-byte ql piece of a super-block (8 lanes per super-block), which carries the low nibbles of scale group
// gi and the high nibbles of group gi + 4 (buf_q6_k.rs:21-48).  6-bit values are rebuilt as bytes for v_dot4; the -32
// offset is applied as -32 * bsum (exact).  off_qh = byte offset of the qh plane = 128 * blocks in the tensor.
template <int R, bool DBG = false>
__device__ __forceinline__ void rows_partial_q6k(const char* __restrict__ w, size_t off_qh, const ActQ8_K& act, int row0, int m,
                                                 int nsb, int lane, float acc[R], int* dbg = nullptr) {
  const size_t n = off_qh / 128;  // blocks in the tensor
  const i32x4* wql = (const i32x4*)w;
  const i32x4* wqh = (const i32x4*)(w + off_qh);
  const char* wsc = w + off_qh + n * 64;
  const unsigned short* wd = (const unsigned short*)(w + off_qh + n * 80);
#pragma unroll
  for (int r = 0; r < R; r++) acc[r] = 0.f;
  const int npieces = nsb * 8;
  for (int c = lane; c < npieces; c += 64) {
    const int sb = c >> 3, h = (c >> 2) & 1, a = (c >> 1) & 1, p = c & 1;
    const int gi = 8 * h + p + 2 * a;  // scale group of the low nibbles; the high nibbles' group is gi + 4
    i32x4 qv[R], hv[R];
    unsigned scw[R];
    unsigned short dw[R];
#pragma unroll
    for (int r = 0; r < R; r++) {
      const int row = row0 + r < m ? row0 + r : m - 1;
      const size_t blk = (size_t)row * nsb + sb;
      qv[r] = __builtin_nontemporal_load(wql + blk * 8 + (c & 7));
      hv[r] = __builtin_nontemporal_load(wqh + blk * 4 + 2 * h + p);
      const signed char* sp = (const signed char*)wsc + blk * 16 + gi;
      scw[r] = (unsigned)(unsigned char)sp[0] | ((unsigned)(unsigned char)sp[4] << 8);
      dw[r] = wd[blk];
    }
    const i32x4* xq = act.q + (size_t)sb * 16 + gi;  // 16 int8 per group
    const i32x4 xl = xq[0], xh = xq[4];
    const float d8 = act.d[sb];
    const short* bs = act.bsums + sb * 16 + gi;
    const int bs_lo = (int)bs[0], bs_hi = (int)bs[4];
#pragma unroll
    for (int r = 0; r < R; r++) {
      int lo = 0, hi = 0;
#pragma unroll
      for (int i = 0; i < 4; i++) {
        const unsigned q = (unsigned)qv[r][i], hb = (unsigned)hv[r][i] >> (2 * a);
        const unsigned ql4 = (q & 0x0F0F0F0Fu) | ((hb & 0x03030303u) << 4);
        const unsigned qh4 = ((q >> 4) & 0x0F0F0F0Fu) | (((hb >> 4) & 0x03030303u) << 4);
        lo = __builtin_amdgcn_sdot4((int)ql4, xl[i], lo, false);
        hi = __builtin_amdgcn_sdot4((int)qh4, xh[i], hi, false);
      }
      lo -= 32 * bs_lo;  // sum (q6 - 32) * q8, exact
      hi -= 32 * bs_hi;
      const int sc_lo = (int)(signed char)(scw[r] & 0xffu), sc_hi = (int)(signed char)(scw[r] >> 8);
      if constexpr (DBG) {  // parity hook: the two scaled group sums of this piece, as the float part consumes them
        dbg[((size_t)r * npieces + c) * 2] = sc_lo * lo;
        dbg[((size_t)r * npieces + c) * 2 + 1] = sc_hi * hi;
      }
      const float dd = h2f(dw[r]) * d8;
      acc[r] += dd * ((float)(sc_lo * lo) + (float)(sc_hi * hi));
    }
  }
}

// R rows of a weight matrix in format FMT against its activation planes (ActQ8_0 for Q4_0 / Q8_0, ActQ8_K for
// Q4_K); `wd` is the format's second plane (f16 scales / 16-byte headers), `nu` the blocks per row
template <int FMT, int R, class ACT>
__device__ __forceinline__ void rows_dot(const i32x4* __restrict__ wq, const unsigned short* __restrict__ wd, const ACT& act,
                                         int row0, int m, int nu, int lane, float acc[R]) {
  if constexpr (FMT == CRABML_HIP_Q4_K)
    rows_partial_q4k<R>(wq, (const i32x4*)wd, act, row0, m, nu, lane, acc);
  else
    rows_partial<FMT, R>(wq, wd, act, row0, m, nu, lane, acc);
}
template <int FMT>
struct ActOf {
  typedef ActQ8_0 type;
};
template <>
struct ActOf<CRABML_HIP_Q4_K> {
  typedef ActQ8_K type;
};
template <>
struct ActOf<CRABML_HIP_Q4_1> {
  typedef ActQ8_1 type;
};

// ---- Q5_0 / Q5_1 / Q2_K / Q3_K: the formats the reference serves with scalar code only --------------------------------------
// (buf_q5_0.rs, buf_q5_1.rs, buf_q2_k.rs, buf_q3_k.rs).  They run the per-op path (matmul_vec, embedding rows, the generic
// decode step); one policy per format describes a lane's 16-byte PIECE of quants: what it loads, the exact integers it yields,
// and the f32 term in the reference's own expression.  Planes (common.hpp): the first plane holds QB bytes of quants per block,
// `off` = n * QB is where the second starts (n = blocks of the tensor), the others follow.
// four bits b -> bit 4 of four bytes (the fifth bit of four 5-bit levels)
__device__ __forceinline__ unsigned spread4_to_bit4(unsigned b) { return ((b * 0x00204081u) & 0x01010101u) << 4; }
__device__ __forceinline__ int quad_sum_i32(int v) {
  v += dpp_i<0xB1>(v);
  v += dpp_i<0x4E>(v);
  return v;
}

struct PieceQ5_0 {  // planes qs[n][16] | qh[n] u32 | d[n] f16; rhs Q8_0
  static constexpr int PIECES = 1, QB = 16, GROUPS = 1;
  typedef ActQ8_0 Act;
  struct W {
    i32x4 q;
    unsigned qh;
    unsigned short d;
  };
  struct X {
    i32x4 x0, x1;
    float dx;
    int xs;
  };
  static __device__ __forceinline__ W load(const char* __restrict__ w, size_t off, size_t n, size_t blk0, int c) {
    const size_t b = blk0 + c;
    W r;
    r.q = __builtin_nontemporal_load((const i32x4*)w + b);
    r.qh = __builtin_nontemporal_load((const unsigned*)(w + off) + b);
    r.d = __builtin_nontemporal_load((const unsigned short*)(w + off + n * 4) + b);
    return r;
  }
  static __device__ __forceinline__ X loadx(const Act& a, int c) { return X{a.q[2 * c], a.q[2 * c + 1], h2f(a.d[c]), a.isum[c]}; }
  // sum (q5 - 16) * q8, exact (buf_q5_0.rs:148-157)
  static __device__ __forceinline__ void ints(const W& w, const X& x, int c, int* o) {
    int s = 0;
#pragma unroll
    for (int i = 0; i < 4; i++) {
      const unsigned v = (unsigned)w.q[i];
      s = __builtin_amdgcn_sdot4((int)((v & 0x0F0F0F0Fu) | spread4_to_bit4((w.qh >> (4 * i)) & 0xFu)), x.x0[i], s, false);
      s = __builtin_amdgcn_sdot4((int)(((v >> 4) & 0x0F0F0F0Fu) | spread4_to_bit4((w.qh >> (16 + 4 * i)) & 0xFu)), x.x1[i], s, false);
    }
    o[0] = s - 16 * x.xs;
  }
  static __device__ __forceinline__ int group_index(int c, int g) { return c; }
  static __device__ __forceinline__ float term(const W& w, const X& x, int lane) {  // buf_q5_0.rs:158
    int si;
    ints(w, x, 0, &si);
    return ((float)si * h2f(w.d)) * x.dx;
  }
};

struct PieceQ5_1 {  // planes qs[n][16] | (d f16, m f16, qh u32)[n]; rhs Q8_1
  static constexpr int PIECES = 1, QB = 16, GROUPS = 1;
  typedef ActQ8_1 Act;
  struct W {
    i32x4 q;
    i32x2 h;  // d | m << 16, qh
  };
  struct X {
    i32x4 x0, x1;
    unsigned ds;  // the Q8_1 block's raw f16 pair d | s << 16
  };
  static __device__ __forceinline__ W load(const char* __restrict__ w, size_t off, size_t n, size_t blk0, int c) {
    const size_t b = blk0 + c;
    W r;
    r.q = __builtin_nontemporal_load((const i32x4*)w + b);
    r.h = __builtin_nontemporal_load((const i32x2*)(w + off) + b);
    return r;
  }
  static __device__ __forceinline__ X loadx(const Act& a, int c) {
    return X{a.q[2 * c], a.q[2 * c + 1], (unsigned)a.d[c] | ((unsigned)a.s[c] << 16)};
  }
  static __device__ __forceinline__ void ints(const W& w, const X& x, int c, int* o) {  // sum q5 * q8 (buf_q5_1.rs:146-155)
    const unsigned qh = (unsigned)w.h[1];
    int s = 0;
#pragma unroll
    for (int i = 0; i < 4; i++) {
      const unsigned v = (unsigned)w.q[i];
      s = __builtin_amdgcn_sdot4((int)((v & 0x0F0F0F0Fu) | spread4_to_bit4((qh >> (4 * i)) & 0xFu)), x.x0[i], s, false);
      s = __builtin_amdgcn_sdot4((int)(((v >> 4) & 0x0F0F0F0Fu) | spread4_to_bit4((qh >> (16 + 4 * i)) & 0xFu)), x.x1[i], s, false);
    }
    o[0] = s;
  }
  static __device__ __forceinline__ int group_index(int c, int g) { return c; }
  // buf_q5_1.rs:156: sumi as f32 * f16(d_w * d_x) + f16(m * s) -- the two products are f16 * f16 rounded to f16, as Q4_1's
  static __device__ __forceinline__ float term(const W& w, const X& x, int lane) {
    int si;
    ints(w, x, 0, &si);
    const unsigned dm = (unsigned)w.h[0];
    return (float)si * h2f(h_mul((unsigned short)(dm & 0xffffu), (unsigned short)(x.ds & 0xffffu))) +
           h2f(h_mul((unsigned short)(dm >> 16), (unsigned short)(x.ds >> 16)));
  }
};

// The activation side of a K-quant piece that spans four 16-element scale groups (Q2_K, Q3_K): piece j = (half, h) of a
// super-block holds qs bytes 32 half + 16 h .. +16, whose 2-bit field s (shift 2 s) belongs to group g = 8 half + 2 s + h =
// elements 16 g .. 16 g + 16 (buf_q2_k.rs:44-67, buf_q3_k.rs:62-86).
struct KGroupsX {
  i32x4 xq[4];
  int bs[4];
  float d8;
};
__device__ __forceinline__ KGroupsX kgroups_loadx(const ActQ8_K& a, int c) {
  const int sb = c >> 2, half = (c >> 1) & 1, h = c & 1;
  KGroupsX x;
#pragma unroll
  for (int s = 0; s < 4; s++) {
    const int g = 8 * half + 2 * s + h;
    x.xq[s] = a.q[(size_t)sb * 16 + g];
    x.bs[s] = (int)a.bsums[sb * 16 + g];
  }
  x.d8 = a.d[sb];
  return x;
}

struct PieceQ2_K {  // planes qs[n][64] | scales[n][16] | (d f16, dmin f16)[n]; rhs Q8_K
  static constexpr int PIECES = 4, QB = 64, GROUPS = 4;
  typedef ActQ8_K Act;
  typedef KGroupsX X;
  struct W {
    i32x4 qv;
    i32x2 sc;  // the half's eight (scale | min << 4) bytes
    unsigned dm;
  };
  static __device__ __forceinline__ W load(const char* __restrict__ w, size_t off, size_t n, size_t blk0, int c) {
    const size_t b = blk0 + (c >> 2);
    W r;
    r.qv = __builtin_nontemporal_load((const i32x4*)w + b * 4 + (c & 3));
    r.sc = __builtin_nontemporal_load((const i32x2*)(w + off) + b * 2 + ((c >> 1) & 1));
    r.dm = __builtin_nontemporal_load((const unsigned*)(w + off + n * 16) + b);
    return r;
  }
  static __device__ __forceinline__ X loadx(const Act& a, int c) { return kgroups_loadx(a, c); }
  static __device__ __forceinline__ unsigned scale_byte(const W& w, int c, int s) {
    return ((unsigned)w.sc[s >> 1] >> (8 * (2 * (s & 1) + (c & 1)))) & 0xFFu;
  }
  static __device__ __forceinline__ int group_dot(const W& w, const X& x, int s) {  // sum q2 * q8 over the group
    int d = 0;
#pragma unroll
    for (int i = 0; i < 4; i++) d = __builtin_amdgcn_sdot4((int)(((unsigned)w.qv[i] >> (2 * s)) & 0x03030303u), x.xq[s][i], d, false);
    return d;
  }
  static __device__ __forceinline__ void ints(const W& w, const X& x, int c, int* o) {
#pragma unroll
    for (int s = 0; s < 4; s++) o[s] = group_dot(w, x, s);
  }
  static __device__ __forceinline__ int group_index(int c, int s) { return (c >> 2) * 16 + 8 * ((c >> 1) & 1) + 2 * s + (c & 1); }
  // buf_q2_k.rs:216-258: the super-block's isum = sum (scale & 15) * group dot and summs = sum bsum * (scale >> 4) are exact
  // integers (the four pieces of a super-block add theirs across the quad), then ONE f32 expression per super-block
  static __device__ __forceinline__ float term(const W& w, const X& x, int c, int lane) {
    int isum = 0, summs = 0;
#pragma unroll
    for (int s = 0; s < 4; s++) {
      const unsigned sc = scale_byte(w, c, s);
      isum += (int)(sc & 15u) * group_dot(w, x, s);
      summs += x.bs[s] * (int)(sc >> 4);
    }
    isum = quad_sum_i32(isum);
    summs = quad_sum_i32(summs);
    const float dall = x.d8 * h2f((unsigned short)(w.dm & 0xffffu)), dmin = x.d8 * h2f((unsigned short)(w.dm >> 16));
    const float t = dall * (float)isum - dmin * (float)summs;
    return (lane & 3) == 0 ? t : 0.0f;
  }
};

struct PieceQ3_K {  // planes qs[n][64] | hmask[n][32] | (scales[12], d f16, 2 unused bytes)[n]; rhs Q8_K
  static constexpr int PIECES = 4, QB = 64, GROUPS = 4;
  typedef ActQ8_K Act;
  typedef KGroupsX X;
  struct W {
    i32x4 qv, hm, sd;
  };
  static __device__ __forceinline__ W load(const char* __restrict__ w, size_t off, size_t n, size_t blk0, int c) {
    const size_t b = blk0 + (c >> 2);
    W r;
    r.qv = __builtin_nontemporal_load((const i32x4*)w + b * 4 + (c & 3));
    r.hm = __builtin_nontemporal_load((const i32x4*)(w + off) + b * 2 + (c & 1));
    r.sd = __builtin_nontemporal_load((const i32x4*)(w + off + n * 32) + b);
    return r;
  }
  static __device__ __forceinline__ X loadx(const Act& a, int c) { return kgroups_loadx(a, c); }
  // sum (q3 - 4) * q8 over group s of the piece: 2 low bits | the hmask bit as bit 2, minus 4 * (sum of the group's q8)
  static __device__ __forceinline__ int group_dot(const W& w, const X& x, int c, int s) {
    const int bit = 4 * ((c >> 1) & 1) + s;
    int d = 0;
#pragma unroll
    for (int i = 0; i < 4; i++) {
      const unsigned u = (((unsigned)w.qv[i] >> (2 * s)) & 0x03030303u) | ((((unsigned)w.hm[i] >> bit) & 0x01010101u) << 2);
      d = __builtin_amdgcn_sdot4((int)u, x.xq[s][i], d, false);
    }
    return d - 4 * x.bs[s];
  }
  // the 6-bit scale of group g = 8 half + 2 s + h, minus 32 (buf_q3_k.rs:44-55: low 4 bits in scales[g & 7]'s low / high nibble,
  // high 2 bits in scales[8 + g % 4] at bit 2 (g / 4))
  static __device__ __forceinline__ int scale(const W& w, int c, int s) {
    const int half = (c >> 1) & 1, j = 2 * s + (c & 1);  // j = g & 7
    const unsigned lo = (((unsigned)w.sd[j >> 2] >> (8 * (j & 3))) >> (4 * half)) & 0xFu;
    const unsigned hi = (((unsigned)w.sd[2] >> (8 * (j & 3))) >> (2 * (2 * half + (s >> 1)))) & 3u;
    return (int)(lo | (hi << 4)) - 32;
  }
  static __device__ __forceinline__ void ints(const W& w, const X& x, int c, int* o) {
#pragma unroll
    for (int s = 0; s < 4; s++) o[s] = group_dot(w, x, c, s);
  }
  static __device__ __forceinline__ int group_index(int c, int s) { return (c >> 2) * 16 + 8 * ((c >> 1) & 1) + 2 * s + (c & 1); }
  // buf_q3_k.rs:303-327 keeps eight i32 lanes per super-block (element e feeds lane e % 8) and multiplies each by d: the fast
  // path adds the lanes first (exact) and multiplies once -- the strict-order kernel keeps the eight lanes
  static __device__ __forceinline__ float term(const W& w, const X& x, int c, int lane) {
    int tot = 0;
#pragma unroll
    for (int s = 0; s < 4; s++) tot += scale(w, c, s) * group_dot(w, x, c, s);
    tot = quad_sum_i32(tot);
    const float t = (h2f((unsigned short)((unsigned)w.sd[3] & 0xffffu)) * x.d8) * (float)tot;
    return (lane & 3) == 0 ? t : 0.0f;
  }
};

// R rows of a matrix in one of those formats against its activation planes; lane l owns pieces l, l + 64, ...
template <class P, int R>
__device__ __forceinline__ void rows_partial_pieces(const char* __restrict__ w, size_t off, size_t n, const typename P::Act& act, int row0,
                                                    int m, int nbr, int lane, float acc[R]) {
#pragma unroll
  for (int r = 0; r < R; r++) acc[r] = 0.f;
  const int np = nbr * P::PIECES;
  for (int c0 = 0; c0 < np; c0 += 64) {
    const int c = c0 + lane;
    const bool live = c < np;  // pieces per row are a multiple of PIECES: a super-block's quad is live or dead as a whole
    const int cc = live ? c : np - 1;
    typename P::W wv[R];
#pragma unroll
    for (int r = 0; r < R; r++) {
      const int row = row0 + r < m ? row0 + r : m - 1;
      wv[r] = P::load(w, off, n, (size_t)row * nbr, cc);
    }
    const typename P::X x = P::loadx(act, cc);
#pragma unroll
    for (int r = 0; r < R; r++) {
      float t;
      if constexpr (P::PIECES == 1)
        t = P::term(wv[r], x, lane);
      else
        t = P::term(wv[r], x, cc, lane);
      acc[r] += live ? t : 0.0f;
    }
  }
}

}  // namespace crabml_hip

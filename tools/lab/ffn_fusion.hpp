// ffn_fusion.hpp -- LAB ARCHIVE (rounds 1-3), not part of the product and not compiled by the build.
// Two cross-edge fusions of the decode step that were built, bit-identical to the launches they replace, and measured SLOWER on
// MI355X (DESIGN.md section 4 "Round 3"; profiles/r03_qkv_tail_ab.log, r03_engine_ab_*.log):
//   QkvTail / sweep_planes_share  the next layer's q/k/v rows as a register-prefetched tail of the ffn_down launch
//   k_ffn                         gate/up + SiLU*mul + quantize + ffn_down + norm as one launch
// They were wired into crabml_amd/csrc/fused.hip (flags CRABML_HIP_LLAMA_QKV_TAIL / _FFN_FUSION) up to commit 3aa434e;
// check that commit out to run them.  Kept here as evidence of what was measured.
#pragma once

// ---- the NEXT layer's q/k/v GEMV as the tail of the ffn_down launch ---------------------------------------------------------
// k_qkv streams 14 MB in 4.6 us on the 8B shape: 2.2 us of stream and a launch's ramp (a GEMV stage costs
// bytes / 6.2 TB/s + 2.6 us).  Its weights do not depend on anything, so the ffn_down launch can hold them: every workgroup
// requests the rows of (up to) 16 (even, odd) row pairs -- 55 KB per workgroup, 4 x 16 bytes + scales per lane -- right before
// its norm hop and keeps them in REGISTERS; the normalized, quantized residual then crosses the workgroups as granules (the
// engine's format: 8 {4 quants, epoch} + 1 {d | aux, epoch} per block, swept by all 16 waves in one round trip: the chip has
// drained its weight stream by then, so the hop is cheap), and the dots run from registers.  Same lane -> block mapping and
// order as k_qkv (rows_partial<FMT, 2>), same epilogue: bit-identical to the separate launch.
struct QkvTail {
  Planes wq, wk, wv;
  QkvEpi e;
  unsigned long long* xq_g;  // dim / 4 quant granules
  unsigned long long* xs_g;  // dim / 32 scale granules
  int pairs_per_wg;          // row pairs per workgroup (<= 16: one per wave)
  int off_d, off_aux;        // act_layout of the dim-sized rhs planes (LDS copy)
};
struct NoQkv {};
template <bool Q>
struct QkvArg {
  typedef NoQkv type;
};
template <>
struct QkvArg<true> {
  typedef QkvTail type;
};
// one wave's share of the sweep of a quantized vector's granules (n / 4 quant + n / 32 scale granules) into LDS planes
// q | d | isum; every load of the share is in flight at once, the share is re-read until each granule carries the epoch
// (bounded: a workgroup that never publishes raises the fault word)
template <int B>
__device__ __forceinline__ void sweep_planes_share(const unsigned long long* qg, const unsigned long long* sg, int n, unsigned epoch,
                                                   unsigned char* P, int off_d, int off_aux, int lane, int part, int nparts, int* fault) {
  unsigned* pq = (unsigned*)P;
  unsigned short* pd = (unsigned short*)(P + off_d);
  int* pa = (int*)(P + off_aux);
  const int nq = n / 4, count = nq + n / 32;
  const int per = ((count + nparts - 1) / nparts + 63) & ~63;
  const int lo = part * per, hi = lo + per < count ? lo + per : count;
  for (int base = lo; base < hi; base += 64 * B) {
    unsigned long long x[B];
    int spins = 0;
    for (;;) {
      bool ok = true;
#pragma unroll
      for (int i = 0; i < B; i++) {
        const int idx = base + i * 64 + lane;
        const int j = idx < hi ? idx : base;
        x[i] = __hip_atomic_load(j < nq ? qg + j : sg + (j - nq), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
#pragma unroll
      for (int i = 0; i < B; i++) ok &= (unsigned)(x[i] >> 32) == epoch;
      if (__all(ok)) break;
      if (++spins > (1 << 19)) {
        if (lane == 0) *fault = 1;
        break;
      }
      __builtin_amdgcn_s_sleep(2);
    }
#pragma unroll
    for (int i = 0; i < B; i++) {
      const int idx = base + i * 64 + lane;
      if (idx < hi) {
        const unsigned v = (unsigned)x[i];
        if (idx < nq) {
          pq[idx] = v;
        } else {
          pd[idx - nq] = (unsigned short)(v & 0xffffu);
          pa[idx - nq] = (int)(short)(v >> 16);
        }
      }
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

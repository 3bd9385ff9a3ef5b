// fused_ffn.hpp -- wo / ffn_down GEMV + residual (+ the next RMSNorm / quantize in the epilogue), gate/up + SiLU*mul, the fused-FFN experiment, greedy sampler
// Part of the fused decode step (fused.hip includes the three fused_*.hpp files once, in order; they are not stand-alone
// translation units: the kernels are launched from fused.hip's host code).
#pragma once
#include "fused_common.hpp"

namespace crabml_hip {
// ---- GEMV + residual: x[row] = W[row].xq + x[row]   (matmul_vec, then add_inplace: arithmetic.rs:27-33) ---
template <int FMT, int R, bool ADD>  // ADD: x[row] += W.xq (residual); else out[row] = W.xq (tensor-parallel partial sum)
__global__ __launch_bounds__(128) void k_gemv_res(Planes w, typename ActOf<FMT>::type act, float* __restrict__ x, int m, int nb) {
  const int lane = threadIdx.x & 63;
  const int wave = blockIdx.x * (blockDim.x >> 6) + wave_in_wg();
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

// strict order: x[row] = (the row's block terms added in block order) + x[row]; 4 waves x 2 rows, dynamic LDS = 8 * nt floats
template <int FMT>
__global__ __launch_bounds__(256) void k_gemv_res_ord(Planes w, typename ActOf<FMT>::type act, float* __restrict__ x, int m, int nb) {
  extern __shared__ __attribute__((aligned(16))) float ord_terms[];
  const int lane = threadIdx.x & 63, wv = wave_in_wg();
  const int wave = blockIdx.x * (blockDim.x >> 6) + wv;
  const int row0 = wave * 2;
  if (row0 >= m) return;
  const float res = (lane < 2 && row0 + lane < m) ? x[row0 + lane] : 0.f;
  const int nt = (nb + 3) & ~3;
  float* T = ord_terms + (size_t)wv * 2 * nt;
  rows_terms<FMT, 2>(w.q, w.d, act, row0, m, nb, lane, T, nt);
  __builtin_amdgcn_wave_barrier();
  if (lane < 2 && row0 + lane < m) x[row0 + lane] = ordered_sum(T + lane * nt, nb) + res;
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

// ---- one-shot all-reduce over peer-mapped inboxes (the production collective) -----------------------------
// Tensor-parallel decode exchanges 2 x dim f32 per layer and token: latency, not bandwidth.  Every rank owns an INBOX
// (device memory, exported with hipIpcGetMemHandle and mapped by its peers with hipIpcOpenMemHandle -- the same mapping
// reaches a peer GPU's HBM over xGMI or another process's buffer on the same GPU): TP_SLOTS slots (0 / 1: the step's
// collectives by segment parity, 2 / 3: host-issued all-reduces, 4 / 5: the vocabulary-split sampler) x nranks rows of `cap`
// granules.  A granule is ONE naturally aligned 8-byte {f32 partial, u32 epoch}: data and tag travel in one
// system-scope store, so there is no flag, no fence and no ordering between a producer's rows (MI355X_MICROARCH.md, "R2
// granule").  Rank r writes its partial row into slot[seg & 1][r] of EVERY peer's inbox and then reads the nranks rows of
// its own inbox, polling a granule until its tag is this step's epoch; the partials are added in rank order
// (p0 + p1) + p2 ..., so every rank computes the same bits (= OracleTpLlamaRunner._sum_in_rank_order).  Two slots are
// enough: a rank cannot finish all-reduce k + 1 before every peer has consumed all-reduce k.
#define TP_SLOTS 6
struct TpP2P {
  unsigned long long* peer[8];  // peer[p] = rank p's inbox as mapped in THIS process (peer[me] = the local allocation)
  int n, me;
  unsigned cap;                 // granules per row
  unsigned salt;                // per decode context (the n-th context created on this group): added to the epoch, so that the
                                // granules a previous context left in the inboxes can never carry a matching tag
  int* fault;
};
__device__ __forceinline__ unsigned long long* tp_row(const TpP2P& t, int owner, int slot, int src) {
  return t.peer[owner] + ((size_t)slot * t.n + src) * t.cap;
}
__device__ __forceinline__ void tp_put(unsigned long long* p, float v, unsigned epoch) {
  __hip_atomic_store(p, ((unsigned long long)epoch << 32) | (unsigned long long)__builtin_bit_cast(unsigned, v), __ATOMIC_RELAXED,
                     __HIP_MEMORY_SCOPE_SYSTEM);
}
__device__ __forceinline__ float tp_get(const TpP2P& t, const unsigned long long* p, unsigned epoch) {
  unsigned long long g = __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
  int tries = 0;
  while ((unsigned)(g >> 32) != epoch && tries < (1 << 22)) {  // bounded (seconds): a peer that never arrives raises a fault, not a hang
    __builtin_amdgcn_s_sleep(4);
    g = __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    tries++;
  }
  if ((unsigned)(g >> 32) != epoch) *t.fault = 2;
  return __builtin_bit_cast(float, (unsigned)g);
}
// kernels that host the collective take a TpP2P by value; every other instantiation takes an empty struct (no kernel
// argument bytes, no code: the single-GPU kernels are exactly what they were)
struct NoTp {};
template <bool TP>
struct TpArg {
  typedef NoTp type;
};
template <>
struct TpArg<true> {
  typedef TpP2P type;
};
// element i of this rank's partial -> the all-reduced value (every lane calls it for its own i; i < cap)
__device__ __forceinline__ float tp_allreduce_elem(const TpP2P& t, float part, int i, unsigned epoch, int slot) {
#pragma unroll
  for (int p = 0; p < 8; p++)  // (static indices into the kernel-argument array; n <= 8 ranks = one xGMI node)
    if (p < t.n && p != t.me) tp_put(tp_row(t, p, slot, t.me) + i, part, epoch);
  float sum = 0.f;
  unsigned long long* mine = nullptr;
#pragma unroll
  for (int p = 0; p < 8; p++)
    if (p == t.me) mine = t.peer[p];
#pragma unroll
  for (int s = 0; s < 8; s++) {
    if (s >= t.n) break;
    const float v = s == t.me ? part : tp_get(t, mine + ((size_t)slot * t.n + s) * t.cap + i, epoch);
    sum = s == 0 ? v : sum + v;
  }
  return sum;
}
// the collective as its own launch (replaces ncclAllReduce on the per-op segment path and in crabml_hip_tp_all_reduce):
// buf[i] <- sum over ranks, in place.  epoch_d != NULL: epoch = *epoch_d * nseg + seg + 1 (decode step, graph-safe)
// slot_base: 0 for the decode step's collectives (slots 0 / 1 by segment parity), 2 for crabml_hip_tp_all_reduce issued by the
// host (slots 2 / 3 by call parity): a host collective between two steps can then never land in the slot a step's first segment
// polls next (round-2 review finding); slots 4 / 5 belong to the vocabulary-split sampler (k_argmax_step_tp)
__global__ __launch_bounds__(256) void k_tp_allreduce(float* __restrict__ buf, int n, TpP2P t, const int* __restrict__ serial_d, int nseg, int seg,
                                                      unsigned epoch_host, int slot_base) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const unsigned epoch = serial_d ? (unsigned)(*serial_d) * (unsigned)nseg + (unsigned)seg + 1u + t.salt : epoch_host;
  buf[i] = tp_allreduce_elem(t, buf[i], i, epoch, slot_base + (seg & 1));
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
// cm: the quants go to LDS class-major (the rhs of Q4_K rows; common.hpp) instead of in element order (a Q6_K matrix of a *_K_M mix)
__device__ __forceinline__ void stage_quant_q8k(const float* __restrict__ x, int nsb, unsigned* sq, float* sd, short* sbs, bool cm) {
  const int lane = threadIdx.x & 63, wave = wave_in_wg(), nw = blockDim.x >> 6;
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
      if (cm)
        q8k_store_class_major((signed char*)sq + sb * 256, lane, o.packed);
      else
        sq[sb * 64 + lane] = o.packed;
      if ((lane & 3) == 0) sbs[sb * 16 + (lane >> 2)] = (short)o.quad_sum;
      if (lane == 0) sd[sb] = o.d;
    }
  }
  __syncthreads();
}

// the finished Q8_K planes of a vector (written by the kernel that produced it: q8k_exchange_store) copied into LDS:
// coalesced 16-byte pieces, one round trip
__device__ __forceinline__ void stage_copy_q8k(const ActQ8_K& act, int nsb, i32x4* sq, float* sd, short* sbs, bool cm) {
  const i32x4* src = cm ? act.qp : act.q;
  for (int i = threadIdx.x; i < nsb * 16; i += blockDim.x) sq[i] = src[i];
  for (int i = threadIdx.x; i < nsb; i += blockDim.x) sd[i] = act.d[i];
  for (int i = threadIdx.x; i < nsb * 2; i += blockDim.x) ((i32x4*)sbs)[i] = ((const i32x4*)act.bsums)[i];
  __syncthreads();
}

// Q8_0 quantizer of an f32 vector into LDS planes (q[k] | d[nb] f16 | isum[nb] i32): a half-wave per 32-element block
// (quant_lane32 = buf_q8_0.rs:87-134, the arithmetic of the gate/up epilogue and of the quantizer launch), four rounds of loads in
// flight per wave.  A tensor-parallel rank's ffn_down runs it as its prologue on h (k_gateup_h leaves h as f32), each workgroup
// for itself: 14 KB of L2 reads at the 70B / 8 shape, under the first weight requests.
// (two phases so that the caller can put its first weight requests BETWEEN them: loads return in order, and the rows of h are what
// the barrier below waits for -- requested behind 64 KB of weights they arrived 2 us later)
struct StageQ8Regs {
  float v[4];
};
__device__ __forceinline__ StageQ8Regs stage_quant_q8_0_request(const float* __restrict__ x, int nb, int b0) {
  const int lane = threadIdx.x & 63, nw = blockDim.x >> 6;
  const int l32 = lane & 31, hf = lane >> 5;
  StageQ8Regs r;
#pragma unroll
  for (int u = 0; u < 4; u++) {
    const int b = b0 + u * 2 * nw + hf;
    r.v[u] = x[(size_t)(b < nb ? b : nb - 1) * 32 + l32];
  }
  return r;
}
__device__ __forceinline__ void stage_quant_q8_0_store(const StageQ8Regs& r, int nb, int b0, signed char* sq, unsigned short* sd, int* ss) {
  const int lane = threadIdx.x & 63, nw = blockDim.x >> 6;
  const int l32 = lane & 31, hf = lane >> 5;
#pragma unroll
  for (int u = 0; u < 4; u++) {
    if (b0 + u * 2 * nw >= nb) break;  // wave-uniform
    const int b = b0 + u * 2 * nw + hf;
    const QLane o = quant_lane32<false>(r.v[u], true);
    if (b < nb) {
      sq[b * 32 + l32] = o.q;
      if (l32 == 0) {
        sd[b] = o.d;
        ss[b] = o.aux;
      }
    }
  }
}
__host__ __device__ inline size_t q8_0_lds_bytes(int nb) { return (size_t)nb * 32 + (size_t)((nb + 1) & ~1) * 2 + (size_t)nb * 4; }

struct NormGather {
  unsigned long long* slots;  // dim/16 granules: each workgroup's ordered sum of squares over its rows
  unsigned long long* pair;   // dim row granules (read by a split chunk's partner / a Q8_K super-block's neighbours)
  const int* serial;          // decode-step serial number (never reset): makes the epoch unique per launch
  int* fault;
  int nseg, seg;
  float* sums;                // DEFER launches: the chunk sums of squares go here as plain floats (read by the NEXT launch)
  signed char* qp = nullptr;  // Q8_K output: the class-major copy of the quants (the plane the next Q4_K GEMV reads)
};
__device__ __forceinline__ unsigned long long ld_granule(const unsigned long long* p) {
  return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
// SPLIT workgroups share one 32-row chunk (16 waves x 2 / SPLIT rows); QIN (Q4_K): the rhs is the f32 vector xin
// The tail of the wo / ffn_down kernels (k_gemv_res_nq, k_ffn): acc[] = this wave's RW row dots.  Publishes the
// workgroup's rows / sum of squares, takes the one in-launch hop, normalizes + quantizes the rows it owns.
// wg_index / nwg_all: this workgroup's index among the SPLIT * nchunks workgroups of the stage.
// DEFER (the fast step's hop-free norm, gemv_core.hpp `RmsTail`; Q8_0 rhs, one workgroup per chunk): no gather.  The workgroup
// quantizes x * w_norm of its own 32 rows and leaves its chunk's sum of squares for the consuming launch.
// ORD (strict-order device, Q4_K): the caller has already put the rows' ORDERED dots into hv[part * ROWS ..] (wave 0, after a barrier);
// the chunk's sum of squares is one 32-element scan from -0.0 (rms_norm.rs:35-38; with two workgroups per chunk the second continues
// the first one's scan from its granule) and the chunk sums are added strictly in chunk order (rms_norm.rs:38-40), as
// k_gemv_res_nq_ord does.
template <int FMT, int SPLIT, bool TP = false, bool DEFER = false, bool ORD = false>
__device__ __forceinline__ void nq_epilogue(float (&acc)[2 / SPLIT], float res, float wn, f32x4 wn4, unsigned epoch, float* hv,
                                            float* __restrict__ x, signed char* __restrict__ q, void* __restrict__ d,
                                            void* __restrict__ isum, const NormGather& ng, float eps, int blk, int part, int nchunks,
                                            int row, int lane, int wave, int wg_index, int nwg_all,
                                            const typename TpArg<TP>::type& tp = typename TpArg<TP>::type{}) {
  constexpr bool Q81 = FMT == CRABML_HIP_Q4_1;
  constexpr bool KQ = FMT == CRABML_HIP_Q4_K;
  constexpr int RW = 2 / SPLIT;
  constexpr int ROWS = 32 / SPLIT;
  // ---- epilogue: publish, one in-launch hop, normalize + quantize -------------------------------------------
  // every row goes out as a {value, epoch} granule when another workgroup needs it (the partner of a split chunk;
  // the seven neighbours of a Q8_K super-block), the workgroup's ordered sum of squares as one more
  constexpr bool ROWG = SPLIT > 1 || KQ;
  static_assert(!ORD || (KQ && !TP && !DEFER), "the ordered epilogue: Q4_K layers on one device");
  if constexpr (!ORD) {
#pragma unroll
    for (int r = 0; r < RW; r++) {
      const float s = wave_sum_f32(acc[r]);
      if (lane == 0) hv[part * ROWS + wave * RW + r] = s;
    }
  }
  __syncthreads();
  if (wave != 0) return;
  // wave 0 owns the stores: ROWS consecutive rows per instruction (x and the row granules are one or two lines,
  // not 32 separate partial writes from 16 waves)
  if (lane < ROWS) {
    float mo = hv[part * ROWS + lane];
    // tensor parallel: this rank's rows are PARTIAL sums over its k slice -- exchange them with the peers' (one-shot
    // all-reduce through the inboxes, rank-order sum) before the residual is added; every rank then holds the same x rows
    if constexpr (TP) mo = tp_allreduce_elem(tp, mo, row + lane, epoch + tp.salt, ng.seg & 1);
    const float xv = mo + res;  // x = matmul_out + x (llama2.rs:266 / :636)
    x[row + lane] = xv;
    hv[part * ROWS + lane] = xv;
    if (ROWG && q != nullptr)
      __hip_atomic_store(ng.pair + row + lane, ((unsigned long long)epoch << 32) | (unsigned long long)__builtin_bit_cast(unsigned, xv),
                         __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
  __builtin_amdgcn_wave_barrier();
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
  // sum of squares of a chunk = (rows 0..15 in order) + (rows 16..31 in order): a split chunk's two workgroups
  // each own one half (norm_quant_block<HALF> computes the same)
  float cs;
  if constexpr (ORD) {
    cs = -0.0f;
    if (SPLIT > 1 && part > 0) cs = poll(ng.slots + 2 * blk);  // the first half's scan, continued
#pragma unroll
    for (int j = 0; j < ROWS / 4; j++) {
      const f32x4 t = ((const f32x4*)hv)[part * (ROWS / 4) + j];
      cs += t[0] * t[0];
      cs += t[1] * t[1];
      cs += t[2] * t[2];
      cs += t[3] * t[3];
    }
  } else {
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
  // q == nullptr (the fast Q4_K step's wo): x and the chunk's sum of squares (a plain store, read after the kernel boundary) are all
  // this launch leaves -- the consuming gate | up launch normalizes and quantizes the row itself (k_gateup_k_lds<.., NORMIN>): no row
  // granules, no gather of the sums, no super-block exchange
  if (q == nullptr) {
    if (lane == 0) ng.sums[wg_index] = cs;
    return;
  }
  if constexpr (DEFER) {
    static_assert(!DEFER || (!KQ && !Q81 && !TP), "the hop-free epilogue: Q8_0 rhs");
    // the chunk's sum of squares for the consuming launch -- a plain store, read after the kernel boundary
    if (SPLIT == 1 && lane == 0) ng.sums[blk] = cs;
    float v;
    if (SPLIT > 1) {
      // the block's other 16 rows live in the partner workgroup: ONE pairwise hand-off (its row granules), no gather of the row
      __builtin_amdgcn_s_waitcnt(0x0F70);  // vmcnt(0): our own granules are on their way before the polls queue up behind them
      const int l32d = lane & 31;
      const bool ownd = l32d >= part * ROWS && l32d < (part + 1) * ROWS;
      v = 0.0f;
      if (lane < 32) {
        if (ownd) {
          v = hv[l32d];
        } else {
          const unsigned long long* p = ng.pair + blk * 32 + l32d;
          unsigned long long g = ld_granule(p);
          int tries = 0;
          while ((unsigned)(g >> 32) != epoch && tries < (1 << 21)) {
            __builtin_amdgcn_s_sleep(1);
            g = ld_granule(p);
            tries++;
          }
          if ((unsigned)(g >> 32) != epoch) *ng.fault = 1;
          v = __builtin_bit_cast(float, (unsigned)g);
        }
      }
      v = __shfl(v, lane & 31, 64);  // (the upper half quantizes a copy, as everywhere)
      // both halves hold the block's 32 rows now: ONE sum of squares per chunk (every consuming wave reads all of them: half
      // as many is 512 bytes less per wave), the same tree in both workgroups; part 0 stores it
      {
        // (the cross-lane reads stay OUTSIDE the lane-0 branch: the compiler sinks the last add of the tree into a branch that only
        // lane 0 executes, and v_readlane then picks up lane 16's unfinished value)
        const float sq = row16_sum_f32(v * v);
        const float tot = rl_f(sq, 0) + rl_f(sq, 16);
        if (lane == 0 && part == 0) ng.sums[blk] = tot;
      }
      const QLane o = quant_lane32<false>(v * wn, true);
      if (lane < 32 && ownd) {
        q[blk * 32 + lane] = o.q;
        if (l32d == part * ROWS) {  // (both halves computed the same block scale / sum: part 0 stores them)
          if (part == 0) {
            ((unsigned short*)d)[blk] = o.d;
            ((int*)isum)[blk] = o.aux;
          }
        }
      }
    } else {
      const QLane o = quant_lane32<false>(hv[lane & 31] * wn, true);
      if (lane < 32) {
        q[blk * 32 + lane] = o.q;
        if (lane == 0) {
          ((unsigned short*)d)[blk] = o.d;
          ((int*)isum)[blk] = o.aux;
        }
      }
    }
    return;
  }
  if (lane == 0)
    __hip_atomic_store(ng.slots + wg_index, ((unsigned long long)epoch << 32) | (unsigned long long)__builtin_bit_cast(unsigned, cs),
                       __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  __builtin_amdgcn_s_waitcnt(0x0F70);  // vmcnt(0): the granule is on its way before the polls queue up behind it
  // the rows of other workgroups first (published before their sums; the loads fly while the stragglers arrive) ...
  // (Round 3 measured the alternative -- every granule of the hop requested at once, stale ones polled afterwards: ffn_down
  // 9.5 -> 10.3 us, profiles/r03_batched_epilogue_polls_ab.log.  Requested early, most granules come back stale and are
  // fetched twice; polled in this order, the wait for the first straggler covers the arrival of the rest.)
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
  // (fast-mode order, shared with norm_quant_block: 64 chunks per round through the DPP tree, rounds added in order)
  float sum = 0.0f;
  const int nwg = nwg_all;
  for (int base = 0; base < nchunks; base += 64) {
    const int c = base + lane;  // this lane's chunk
    float cv;
    if constexpr (ORD) {  // (SPLIT = 2: the second workgroup's granule holds the whole chunk's scan)
      cv = c < nchunks ? poll(ng.slots + (SPLIT > 1 ? 2 * c + 1 : c)) : 0.0f;
#pragma unroll
      for (int i = 0; i < 64; i++) sum += rl_f(cv, i);
      continue;
    }
    if (SPLIT > 1) {
      const float h0 = c < nchunks ? poll(ng.slots + 2 * c) : 0.0f;
      const float h1 = c < nchunks ? poll(ng.slots + 2 * c + 1) : 0.0f;
      cv = h0 + h1;  // chunk = its two halves
    } else {
      cv = c < nchunks ? poll(ng.slots + c) : 0.0f;
    }
    sum += wave_sum_f32(cv);  // lanes past the grid add +0.0
  }
  (void)nwg;
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
      q8k_store_class_major(ng.qp + sb * 256, lane, o.packed);
      if ((lane & 3) == 0) ((short*)isum)[sb * 16 + (lane >> 2)] = (short)o.quad_sum;
    }
    if (lane == 0 && (blk & 7) == 0 && part == 0) ((float*)d)[sb] = o.d;
  }
}

// QIN (Q4_K): 0 = the rhs planes are read from global memory; 1 = the rhs arrives as f32 (xin) and is quantized into LDS
// by this workgroup; 2 = the finished planes (act) are copied into LDS
// ORD (strict-order device, Q4_K with QIN 1 / 2): the rows' nine-term records go to LDS behind the rhs planes (q4k_class_terms; dynamic
// LDS = the planes rounded up to 16 bytes + ROWS * q4k_rec_stride(nb) floats), wave 0 adds them in super-block order, the epilogue keeps the
// reference's norm order: every bit equals the per-op launches (k_gemv_exact_q4k + k_norm_f32 + k_quantize_q8_k).
template <int FMT, int SPLIT, int QIN = 0, bool TP = false, bool DEFER = false, bool ORD = false>
__global__ __launch_bounds__(1024) void k_gemv_res_nq(Planes w, typename ActOf<FMT>::type act, const float* __restrict__ xin,
                                                      float* __restrict__ x,
                                                      const float* __restrict__ wnext, float eps,
                                                      signed char* __restrict__ q, void* __restrict__ d,
                                                      void* __restrict__ isum, NormGather ng, int nb, Planes6 w6,
                                                      typename TpArg<TP>::type tp) {
  constexpr bool KQ = FMT == CRABML_HIP_Q4_K;  // Q4_K weights: nb counts super-blocks, the output is Q8_K
  constexpr int RW = 2 / SPLIT;         // rows per wave
  constexpr int ROWS = 32 / SPLIT;      // rows per workgroup
  __shared__ __attribute__((aligned(16))) float hv[32];
  const int lane = threadIdx.x & 63, wave = wave_in_wg();
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
      if constexpr (QIN == 2)
        stage_copy_q8k(act, nb, lds_act, sd, sbs, false);
      else
        stage_quant_q8k(xin, nb, (unsigned*)lds_act, sd, sbs, false);
      const ActQ8_K la6{lds_act, sd, sbs, lds_act};
      if constexpr (ORD) {  // the strict-order step: the rows' records (rows_terms_q6k), chained in super-block order by wave 0
        const int stride = q4k_rec_stride(nb);
        float* T = (float*)((char*)lds_act + (((size_t)nb * 292 + 15) & ~(size_t)15));
        rows_terms_q6k<RW>(w6.base, w6.off_qh, la6, row, nchunks * 32, nb, lane, T + (size_t)(wave * RW) * stride, stride);
        __syncthreads();
        if (wave == 0 && lane < ROWS) hv[part * ROWS + lane] = q4k_ordered_sum(T + (size_t)lane * stride, nb);
        nq_epilogue<FMT, SPLIT, TP, false, true>(acc, res, wn, wn4, epoch, hv, x, q, d, isum, ng, eps, blk, part, nchunks, row, lane, wave,
                                                 (int)blockIdx.x, (int)gridDim.x, tp);
        return;
      }
      rows_partial_q6k<RW>(w6.base, w6.off_qh, la6, row, nchunks * 32, nb, lane, acc);
      nq_epilogue<FMT, SPLIT, TP>(acc, res, wn, wn4, epoch, hv, x, q, d, isum, ng, eps, blk, part, nchunks, row, lane, wave,
                                  (int)blockIdx.x, (int)gridDim.x, tp);
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
    if constexpr (QIN == 2)
      stage_copy_q8k(act, nb, lds_act, sd, sbs, true);
    else
      stage_quant_q8k(xin, nb, (unsigned*)lds_act, sd, sbs, true);
    const ActQ8_K la{lds_act, sd, sbs, lds_act};
    if constexpr (ORD) {
      const int stride = q4k_rec_stride(nb);
      float* T = (float*)((char*)lds_act + (((size_t)nb * 292 + 15) & ~(size_t)15));
      float* Tw = T + (size_t)(wave * RW) * stride;
#pragma unroll
      for (int it = 0; it < PRE; it++) {
        const int c = it * 64 + lane;
        const bool live = c < nb * 8;
        const int cc = live ? c : nb * 8 - 1;
        const Q4KX xx = q4k_loadx(la, cc);
#pragma unroll
        for (int r = 0; r < RW; r++) q4k_class_terms<false>(pw[it][r], xx, cc, live, Tw + (size_t)r * stride + (cc >> 3) * 12);
      }
      rows_terms_q4k<RW, false>(w.q, (const i32x4*)w.d, la, row, nchunks * 32, nb, lane, Tw, stride, PRE * 64);
      __syncthreads();
      if (wave == 0 && lane < ROWS) hv[part * ROWS + lane] = q4k_ordered_sum(T + (size_t)lane * stride, nb);
      nq_epilogue<FMT, SPLIT, TP, false, true>(acc, res, wn, wn4, epoch, hv, x, q, d, isum, ng, eps, blk, part, nchunks, row, lane, wave,
                                               (int)blockIdx.x, (int)gridDim.x, tp);
      return;
    }
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
  } else if constexpr (QIN == 1) {
    // the rhs arrives as f32 (h from k_gateup_h): quantized to Q8_0 into LDS by this workgroup, under the first weight requests
    static_assert(KQ || QIN != 1 || FMT == CRABML_HIP_Q4_0 || FMT == CRABML_HIP_Q8_0, "Q8_0 rhs");
    using F = BlockFmt<FMT>;
    extern __shared__ i32x4 lds_act[];
    signed char* sq = (signed char*)lds_act;
    unsigned short* sd = (unsigned short*)(sq + (size_t)nb * 32);
    int* ss = (int*)(sd + ((nb + 1) & ~1));
#pragma unroll
    for (int r = 0; r < RW; r++) acc[r] = 0.f;
    const int nu = nb * F::UNITS;
    typename F::Blk ka0[RW], kb0[RW];
    const int ua = lane < nu ? lane : nu - 1, ub = lane + 64 < nu ? lane + 64 : nu - 1;
    const int nw16 = (int)(blockDim.x >> 6);
    StageQ8Regs hq = stage_quant_q8_0_request(xin, nb, wave * 2);  // h first ...
#pragma unroll
    for (int r = 0; r < RW; r++) {                                  // ... the weights behind it, in flight while h is quantized
      ka0[r] = F::load(w.q, w.d, (size_t)(row + r), nb, ua);
      kb0[r] = F::load(w.q, w.d, (size_t)(row + r), nb, ub);
    }
    for (int b0 = wave * 2; b0 < nb; b0 += 8 * nw16) {
      if (b0 != wave * 2) hq = stage_quant_q8_0_request(xin, nb, b0);
      stage_quant_q8_0_store(hq, nb, b0, sq, sd, ss);
    }
    __syncthreads();
    const ActQ8_0 la{(const i32x4*)sq, sd, ss};
    if (lane < nu) {
      const XUnit xa = F::loadx(la, ua);
#pragma unroll
      for (int r = 0; r < RW; r++) acc[r] += F::term(ka0[r], xa);
    }
    if (lane + 64 < nu) {
      const XUnit xb = F::loadx(la, ub);
#pragma unroll
      for (int r = 0; r < RW; r++) acc[r] += F::term(kb0[r], xb);
    }
    for (int u = lane + 128; u < nu; u += 128) {
      const int u2 = u + 64;
      const bool two = u2 < nu;
      const int uu = two ? u2 : u;
      typename F::Blk ka[RW], kb[RW];
#pragma unroll
      for (int r = 0; r < RW; r++) {
        ka[r] = F::load(w.q, w.d, (size_t)(row + r), nb, u);
        kb[r] = F::load(w.q, w.d, (size_t)(row + r), nb, uu);
      }
      const XUnit xa = F::loadx(la, u), xb = F::loadx(la, uu);
#pragma unroll
      for (int r = 0; r < RW; r++) acc[r] += F::term(ka[r], xa);
      if (two) {
#pragma unroll
        for (int r = 0; r < RW; r++) acc[r] += F::term(kb[r], xb);
      }
    }
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
  nq_epilogue<FMT, SPLIT, TP, DEFER>(acc, res, wn, wn4, epoch, hv, x, q, d, isum, ng, eps, blk, part, nchunks, row, lane, wave,
                                     (int)blockIdx.x, (int)gridDim.x, tp);
}

// ---- strict order: wo / ffn_down + residual + the next RMSNorm + quantize in ONE launch, every sum in the reference's order ----------
// k_gemv_res_nq's structure (32-row chunks, SPLIT workgroups per chunk, one in-launch hop) with:
//   * the GEMV rows from block terms added in block order (rows_terms / ordered_sum; term table ROWS x (nt + 4) floats of dynamic LDS);
//   * a chunk's sum of squares as ONE 32-element scan (rms_norm.rs:35-38): with two workgroups per chunk the second continues the
//     first one's scan from its granule (one more dependent round trip, for that half of the workgroups);
//   * the chunk sums added strictly in chunk order through v_readlane (rms_norm.rs:38-40), as norm_quant_block does with half = 0.
// Same outputs as k_gemv_res_ord + k_norm_quant(half = 0), bit for bit, in one launch instead of two.
template <int FMT, int SPLIT, bool PIPE>  // PIPE: the chain follows the stream step by step (below); else one chain behind the last load
__global__ __launch_bounds__(1024) void k_gemv_res_nq_ord(Planes w, typename ActOf<FMT>::type act, float* __restrict__ x,
                                                          const float* __restrict__ wnext, float eps, signed char* __restrict__ q,
                                                          void* __restrict__ d, void* __restrict__ isum, NormGather ng, int nb) {
  constexpr bool Q81 = FMT == CRABML_HIP_Q4_1;
  constexpr int RW = 2 / SPLIT, ROWS = 32 / SPLIT;
  extern __shared__ __attribute__((aligned(16))) float ord_terms[];
  __shared__ __attribute__((aligned(16))) float hv[32];
  const int lane = threadIdx.x & 63, wave = wave_in_wg();
  const int blk = blockIdx.x / SPLIT, part = blockIdx.x % SPLIT;
  const int nchunks = gridDim.x / SPLIT;
  const int row_wg = blk * 32 + part * ROWS;
  float res = 0.f, wn = 0.f;
  unsigned epoch = 0;
  if (wave == 0) {
    if (lane < ROWS) res = x[row_wg + lane];
    wn = wnext[blk * 32 + (lane & 31)];
    epoch = (unsigned)(*ng.serial) * (unsigned)ng.nseg + (unsigned)ng.seg + 1u;
  }
  const int nt = ((nb + 3) & ~3) + 4;
  // The rows' block terms into the table, step by step (64 units per row); each wave posts how many steps of ITS rows are in the
  // table, and wave 0 -- between its own steps, while its next loads are in flight -- lets lane r add what has arrived of row r, in
  // block order.  The chain (nb dependent adds per row: 448 for ffn_down) thus runs under the stream instead of behind its last load.
  __shared__ int prog[16];
  if (PIPE) {
    if (lane == 0) prog[wave] = 0;
    __syncthreads();
  }
  using F = BlockFmt<FMT>;
  constexpr int TPS = 64 / F::UNITS;  // terms per row and step
  const int nu = nb * F::UNITS;
  float csum = 0.0f;                  // wave 0, lane r < ROWS: the running sum of row r
  int cst = 0;                        // whole steps of every row already added (wave-uniform)
  const float* trow = ord_terms + (size_t)(lane < ROWS ? lane : 0) * nt;
  auto chain_step = [&](int st) {     // the TPS terms of step st, every row at once: all reads first, then the dependent adds
    const float* t = trow + st * TPS;
    f32x4 v[TPS / 4];
#pragma unroll
    for (int i = 0; i < TPS / 4; i++) v[i] = *(const f32x4*)(t + 4 * i);
#pragma unroll
    for (int i = 0; i < TPS / 4; i++) {
      csum += v[i][0];
      csum += v[i][1];
      csum += v[i][2];
      csum += v[i][3];
    }
  };
  if constexpr (!PIPE) {
    rows_terms<FMT, RW>(w.q, w.d, act, row_wg + wave * RW, 0x7fffffff, nb, lane, ord_terms + (size_t)(wave * RW) * nt, nt);
  } else {
    float* T = ord_terms + (size_t)(wave * RW) * nt;
    const int row = row_wg + wave * RW;
    typename F::Blk cur[RW], nxt[RW];
    XUnit xc, xn;
    auto fetch = [&](int u0, typename F::Blk (&blk_)[RW], XUnit& xx) {
      const int u = u0 + lane;
      const int uu = u < nu ? u : nu - 1;
#pragma unroll
      for (int r = 0; r < RW; r++) blk_[r] = F::load(w.q, w.d, (size_t)(row + r), nb, uu);
      xx = F::loadx(act, uu);
    };
    fetch(0, cur, xc);
    int step = 0;
    for (int u0 = 0; u0 < nu; u0 += 64, step++) {
      // (unconditional, clamped: with a conditional request the compiler waits for every load in flight before each use)
      fetch(u0 + 64 < nu ? u0 + 64 : u0, nxt, xn);
      const int u = u0 + lane;
      const bool live = u < nu;
#pragma unroll
      for (int r = 0; r < RW; r++) {
        const float t = F::term(cur[r], xc);
        if (live && (F::UNITS == 1 || (lane & 1) == 0)) T[r * nt + u / F::UNITS] = t;
      }
      // (LDS executes a wave's instructions in order: the flag store lands after the term stores, and a reader that has the flag
      // issues its term loads after it.  A workgroup-scope release here would also wait for the next step's global loads.)
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      if (lane == 0) __hip_atomic_store(&prog[wave], step + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
      if (wave == 0) {  // the steps EVERY wave has posted (whole steps only; the ragged last one waits for the barrier)
        int pr = __hip_atomic_load(&prog[lane & 15], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        pr = min(pr, dpp_i<0xB1>(pr));
        pr = min(pr, dpp_i<0x4E>(pr));
        pr = min(pr, dpp_i<0x141>(pr));
        pr = min(pr, dpp_i<0x140>(pr));
        const int ready = min(__builtin_amdgcn_readlane(pr, 0), nu / 64);
        while (cst < ready) {
          if (lane < ROWS) chain_step(cst);
          cst++;
        }
      }
#pragma unroll
      for (int r = 0; r < RW; r++) cur[r] = nxt[r];
      xc = xn;
    }
  }
  __syncthreads();
  if (wave != 0) return;
  if (lane < ROWS) {
    if constexpr (PIPE) {
      for (; cst < nu / 64; cst++) chain_step(cst);            // what arrived after wave 0's last look
      for (int i = cst * TPS; i < nb; i++) csum += trow[i];    // (a ragged last step)
    } else {
      csum = ordered_sum(trow, nb);
    }
    const float xv = csum + res;                               // x = matmul_out + x (llama2.rs:266 / :636)
    x[row_wg + lane] = xv;
    hv[part * ROWS + lane] = xv;
    if (SPLIT > 1)
      __hip_atomic_store(ng.pair + row_wg + lane, ((unsigned long long)epoch << 32) | (unsigned long long)__builtin_bit_cast(unsigned, xv),
                         __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
  __builtin_amdgcn_wave_barrier();
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
  // the chunk's sum of squares: one scan over its 32 rows, starting from -0.0 as norm_quant_block does
  float cs = -0.0f;
  if (SPLIT > 1 && part > 0) cs = poll(ng.slots + 2 * blk);  // the first half's scan, continued
#pragma unroll
  for (int j = 0; j < ROWS / 4; j++) {
    const f32x4 t = ((const f32x4*)hv)[part * (ROWS / 4) + j];
    cs += t[0] * t[0];
    cs += t[1] * t[1];
    cs += t[2] * t[2];
    cs += t[3] * t[3];
  }
  if (lane == 0)
    __hip_atomic_store(ng.slots + blockIdx.x, ((unsigned long long)epoch << 32) | (unsigned long long)__builtin_bit_cast(unsigned, cs),
                       __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  __builtin_amdgcn_s_waitcnt(0x0F70);  // vmcnt(0)
  const int l32 = lane & 31;
  const bool own = l32 >= part * ROWS && l32 < (part + 1) * ROWS;
  float v = 0.0f;
  if (SPLIT > 1) {
    if (lane < 32) v = own ? hv[l32] : poll(ng.pair + blk * 32 + l32);
  } else {
    v = hv[l32];
  }
  // every chunk's sum (SPLIT = 2: the second workgroup's granule holds the whole chunk), added in chunk order
  float sum = 0.0f;
  for (int base = 0; base < nchunks; base += 64) {
    const int c = base + lane;
    const float cv = c < nchunks ? poll(ng.slots + (SPLIT > 1 ? 2 * c + 1 : c)) : 0.0f;
#pragma unroll
    for (int i = 0; i < 64; i++) sum += rl_f(cv, i);
  }
  const float rms = sqrtf(sum / (float)(nchunks * 32) + eps);
  const float xn = (v / rms) * wn;
  const QLane o = quant_lane32<Q81>(xn, true);
  if (lane < 32 && own) {
    q[blk * 32 + lane] = o.q;
    if (lane == 0) {
      ((unsigned short*)d)[blk] = o.d;
      store_qaux<Q81>(isum, blk, o.aux);
    }
  }
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
  const int row = blockIdx.x * (blockDim.x >> 6) + wave_in_wg();
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
// QOUT: h leaves the kernel as Q8_K planes (the rhs of ffn_down) as well: the eight workgroups of a 256-row super-block exchange
// their rows as granules (q8k_exchange_store); hidden % 256 == 0.
// ORD (strict-order device): the nine-term records of the 32 gate and 32 up rows go to LDS behind the planes (dynamic LDS = the planes
// rounded up to 16 bytes + 64 * q4k_rec_stride(nsb) floats) and one lane per (matrix, row) adds them in super-block order (q4k_ordered_sum):
// h bit for bit as k_gemv_exact_q4k x 2 + k_gateup_epi leave it; m % 32 == 0.
// NORMIN (the fast step): the rhs arrives as the f32 row xin (wo's output, residual added) with wo's chunk sums of squares (csums:
// sum_parts per 32-row chunk) -- every workgroup adds the sums, normalizes (x / rms) * wnorm and quantizes the row to Q8_K into LDS
// itself, a wave per super-block: 32 KB of L2 reads per workgroup instead of wo's two in-launch hops (every workgroup of wo waiting
// for all others' sums, then for its super-block's seven neighbours).  The sums keep the order the gathering epilogue used
// (nq_epilogue: a chunk = its halves; 64 chunks per round through wave_sum_f32; rounds added in order), so the planes are bit for
// bit the ones wo used to leave.
template <bool QOUT, bool ORD = false, bool NORMIN = false>
__global__ __launch_bounds__(1024, 8) void k_gateup_k_lds(Planes wg, Planes wu, ActQ8_K act, const unsigned short* __restrict__ exp_tab,
                                                       float* __restrict__ h, int m, int nsb, Q8KExchange ex, signed char* __restrict__ oq,
                                                       float* __restrict__ od, short* __restrict__ obs, signed char* __restrict__ oqp,
                                                       const float* __restrict__ xin, const float* __restrict__ wnorm, float eps,
                                                       const float* __restrict__ csums, int sum_parts) {
  extern __shared__ i32x4 lds_act[];  // q[k] | d[k/256] f32 | bsums[k/16] i16
  const int k = nsb * 256;
  i32x4* sq = lds_act;
  float* sd = (float*)(sq + k / 16);
  short* sbs = (short*)(sd + nsb);
  const int lane = threadIdx.x & 63, wave = wave_in_wg();
  const int row0 = blockIdx.x * 32 + wave * 2;
  // Round 4: the first round of gate pieces is requested BEFORE the activation planes are staged (the weights depend on nothing:
  // their HBM round trip runs under the staging and its barrier); the rest is two rows_partial_q4k passes as before, so every
  // row's pieces are still added in ascending order.  (Measured and not kept: gate and up rows advancing together, four pieces
  // per lane in flight with a second register set -- 90 VGPRs, ONE workgroup per CU, 18.0 us instead of 17.1; pinned to 64
  // VGPRs it spills.)
  const int nch = nsb * 8;
  // NORMIN: the row, the norm weights and wo's chunk sums are requested BEFORE the first weight pieces (loads return in order: behind
  // the pieces they would wait for an HBM round trip instead of an L2 one).  (host: nsb <= 32 -- at most two super-blocks per wave,
  // four rounds of 64 chunks)
  f32x4 xv[2], wv[2];
  float ca[2];
  if constexpr (NORMIN) {
    static_assert(!ORD, "the strict-order step keeps wo's ordered epilogue");
    const int nch32 = k / 32;
    {  // round `wave` of the chunk sums (waves past the last round re-read its last chunk: unused)
      const int cch = wave * 64 + lane, cc = cch < nch32 ? cch : nch32 - 1;
      ca[0] = csums[cc * sum_parts];
      ca[1] = csums[cc * sum_parts + sum_parts - 1];
    }
#pragma unroll
    for (int u = 0; u < 2; u++) {
      const int sb = wave + 16 * u;
      xv[u] = ((const f32x4*)xin)[(sb < nsb ? sb : 0) * 64 + lane];
      wv[u] = ((const f32x4*)wnorm)[(sb < nsb ? sb : 0) * 64 + lane];
    }
  }
  Q4KPiece<false> pw[2];
#pragma unroll
  for (int r = 0; r < 2; r++)
    pw[r] = q4k_load<false>(wg.q, (const i32x4*)wg.d, (size_t)(row0 + r < m ? row0 + r : m - 1), nsb, lane < nch ? lane : nch - 1, lane);
  if constexpr (NORMIN) {
    const int nch32 = k / 32, nrounds = (nch32 + 63) / 64;
    // round r of the chunk sums = chunks 64 r .. + 63, one per lane of wave r (lanes past the row add +0.0, as the gather did), rounds
    // added in order
    __shared__ float s_round[16];
    if (wave < nrounds) {
      const int cch = wave * 64 + lane;
      const float cs = cch < nch32 ? (sum_parts == 2 ? ca[0] + ca[1] : ca[0]) : 0.0f;
      const float ws = wave_sum_f32(cs);
      if (lane == 0) s_round[wave] = ws;
    }
    __syncthreads();
    float sum = 0.0f;
    for (int r = 0; r < nrounds; r++) sum += s_round[r];
    const float rms = sqrtf(sum / (float)k + eps);
#pragma unroll
    for (int u = 0; u < 2; u++) {
      const int sb = wave + 16 * u;
      if (sb >= nsb) break;  // (wave-uniform)
      f32x4 xn;
#pragma unroll
      for (int i = 0; i < 4; i++) xn[i] = (xv[u][i] / rms) * wv[u][i];  // rms_norm.rs:41-45, then the weight (llama2.rs:611)
      const Q8KLane o = q8k_wave_quant(xn, lane);
      q8k_store_class_major((signed char*)sq + sb * 256, lane, o.packed);
      if ((lane & 3) == 0) sbs[sb * 16 + (lane >> 2)] = (short)o.quad_sum;
      if (lane == 0) sd[sb] = o.d;
    }
  } else {
    for (int i = threadIdx.x; i < k / 16; i += 1024) sq[i] = act.qp[i];  // (class-major: the rhs of Q4_K rows)
    for (int i = threadIdx.x; i < nsb; i += 1024) sd[i] = act.d[i];
    for (int i = threadIdx.x; i < k / 16; i += 1024) sbs[i] = act.bsums[i];
  }
  __syncthreads();
  const ActQ8_K la{sq, sd, sbs, sq};
  __shared__ float hv[32];
  if constexpr (ORD) {
    const int stride = q4k_rec_stride(nsb);
    float* T = (float*)((char*)lds_act + (((size_t)nsb * 292 + 15) & ~(size_t)15));  // rows 0..31: gate, 32..63: up
    float* Tg = T + (size_t)(wave * 2) * stride;
    {
      const bool live = lane < nch;
      const int cc = live ? lane : nch - 1;
      const Q4KX x = q4k_loadx(la, cc);
#pragma unroll
      for (int r = 0; r < 2; r++) q4k_class_terms<false>(pw[r], x, cc, live, Tg + (size_t)r * stride + (cc >> 3) * 12);
    }
    rows_terms_q4k<2, false>(wg.q, (const i32x4*)wg.d, la, row0, m, nsb, lane, Tg, stride, 64);
    rows_terms_q4k<2, false>(wu.q, (const i32x4*)wu.d, la, row0, m, nsb, lane, T + (size_t)(32 + wave * 2) * stride, stride);
    __syncthreads();
    if (wave == 0) {
      const float s = q4k_ordered_sum(T + (size_t)lane * stride, nsb);  // lane < 32: gate row `lane`; else up row `lane - 32`
      const float u = __shfl(s, (lane & 31) + 32, 64);
      if (lane < 32) {
        const float hval = silu_mul(s, u, exp_tab);
        h[(int)blockIdx.x * 32 + lane] = hval;
        hv[lane] = hval;
      }
      if constexpr (QOUT) {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        __builtin_amdgcn_wave_barrier();
        q8k_exchange_store(ex, hv, (int)blockIdx.x * 32, 32, lane, oq, od, obs, oqp);
      }
    }
    return;
  }
  if (row0 >= m) return;
  // (Round 6, measured and not kept: gate and up rows advancing together with quad-exchanged headers -- four pieces per lane and
  // step, two request rounds instead of four, 48 VGPRs: 17.1 us against 15.8.  Round 4's whole-header form of the same idea: 18.0.)
  float ag[2] = {0.f, 0.f}, au[2];
  if (lane < nch) {
    const Q4KX x = q4k_loadx(la, lane);
#pragma unroll
    for (int r = 0; r < 2; r++) ag[r] += q4k_term<false>(pw[r], x, lane);
  }
  rows_partial_q4k<2, false>(wg.q, (const i32x4*)wg.d, la, row0, m, nsb, lane, ag, 64);
  rows_partial_q4k<2, false>(wu.q, (const i32x4*)wu.d, la, row0, m, nsb, lane, au);
#pragma unroll
  for (int r = 0; r < 2; r++) {
    const float g = wave_sum_f32(ag[r]), u = wave_sum_f32(au[r]);
    if (lane == 0 && row0 + r < m) {
      const float hval = silu_mul(g, u, exp_tab);
      h[row0 + r] = hval;
      if (QOUT) hv[wave * 2 + r] = hval;
    }
  }
  if constexpr (QOUT) {
    __syncthreads();
    if (wave == 0) q8k_exchange_store(ex, hv, (int)blockIdx.x * 32, 32, lane, oq, od, obs, oqp);
  }
}

// Same, with the Q8_0 quantization of h (the rhs of ffn_down) folded in: a 1024-thread workgroup owns 32
// consecutive hidden rows = one quant block; each of its 16 waves computes 2 rows (4 weight rows in flight),
// parks the h values in LDS, and one half-wave quantizes the block (buf_q8_0.rs:87-134).  hidden/32
// workgroups (448 for Llama-3-8B) are all resident at once (2 per CU).  Saves a launch per layer.
// (Round 3: two units per row in flight -- 8 weight loads per lane and step -- measured slower on the full grid, 12.9 -> 14.6 us,
// and on a tensor-parallel rank's 112 workgroups, 10.65 -> 11.35 us: a CU's ~26 GB/s is not a matter of bytes in flight.)
// DEFER: the rhs planes come from a hop-free wo launch -- the row dots are multiplied by 1 / rms (RmsTail, gemv_core.hpp)
template <int FMT, bool DEFER = false>
__global__ __launch_bounds__(1024) void k_gateup_q(Planes wg, Planes wu, typename ActOf<FMT>::type act,
                                                   const unsigned short* __restrict__ exp_tab, signed char* __restrict__ q,
                                                   unsigned short* __restrict__ d, void* __restrict__ isum, int nb, RmsTail rt) {
  using F = BlockFmt<FMT>;
  constexpr bool Q81 = FMT == CRABML_HIP_Q4_1;
  __shared__ float hv[32];
  const int lane = threadIdx.x & 63, wave = wave_in_wg();
  const int blk = blockIdx.x;
  const int row = blk * 32 + wave * 2;  // rows row, row+1
  RmsReq rq{0.f, 0.f};
  if constexpr (DEFER) rq = rms_request(rt, lane);
  float g0 = 0.f, g1 = 0.f, u0 = 0.f, u1 = 0.f;
  const int nu = nb * F::UNITS;
  float inv_rms = 1.0f;
  if constexpr (DEFER) {
    // a uniform trip count (lanes past the row's units redo the last one and add nothing), so that the whole wave can reduce the
    // chunk sums INSIDE the first step -- behind its weight requests, while they are in flight: everything a wave does after its
    // last step sits on the launch's critical path (all workgroups are resident at once and finish together)
    for (int u0i = 0; u0i < nu; u0i += 64) {
      const int u = u0i + lane;
      const bool live = u < nu;
      const int uu = live ? u : nu - 1;
      typename F::Blk bg0 = F::load(wg.q, wg.d, (size_t)row, nb, uu);
      typename F::Blk bu0 = F::load(wu.q, wu.d, (size_t)row, nb, uu);
      typename F::Blk bg1 = F::load(wg.q, wg.d, (size_t)row + 1, nb, uu);
      typename F::Blk bu1 = F::load(wu.q, wu.d, (size_t)row + 1, nb, uu);
      const XUnit x = F::loadx(act, uu);
      if (u0i == 0) {
        // (pinned behind this step's requests: the empty asm keeps the reduction from being hoisted out of the loop, the scheduling
        // barrier from being moved above the loads)
        __builtin_amdgcn_sched_barrier(0);
        asm volatile("" : "+v"(rq.v0), "+v"(rq.v1));
        inv_rms = rms_finish(rt, rq, lane);
      }
      const float t0 = F::term(bg0, x), t1 = F::term(bu0, x), t2 = F::term(bg1, x), t3 = F::term(bu1, x);
      g0 += live ? t0 : 0.0f;
      u0 += live ? t1 : 0.0f;
      g1 += live ? t2 : 0.0f;
      u1 += live ? t3 : 0.0f;
    }
  } else {
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
  }
  g0 = wave_sum_f32(g0);
  u0 = wave_sum_f32(u0);
  g1 = wave_sum_f32(g1);
  u1 = wave_sum_f32(u1);
  if constexpr (DEFER) {
    g0 *= inv_rms;
    u0 *= inv_rms;
    g1 *= inv_rms;
    u1 *= inv_rms;
  }
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
// The rows of k_gateup_q with h left as f32 (ffn_down quantizes it in its prologue, stage_quant_q8_0): without the 32-row quant block
// the workgroup's row count is free.  For a tensor-parallel rank, whose hidden / tp / 32 workgroups would cover less than half the
// CUs (112 of 256 at the 70B / 8 shape: 10.8 us for 33 MB, a CU streams ~26 GB/s whatever is resident), the host picks a row
// count that gives one workgroup per CU.  Same dots, same SiLU * mul: h -- and the planes ffn_down makes of it -- bit for bit.
template <int FMT>
__global__ __launch_bounds__(1024) void k_gateup_h(Planes wg, Planes wu, typename ActOf<FMT>::type act, const unsigned short* __restrict__ exp_tab,
                                                   float* __restrict__ h, int m, int nb) {
  using F = BlockFmt<FMT>;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int row = ((int)blockIdx.x * (int)(blockDim.x >> 6) + wave) * 2;
  if (row >= m) return;
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
    h[row] = silu_mul(g0, u0, exp_tab);
    h[row + 1] = silu_mul(g1, u1, exp_tab);
  }
}
// strict order: k_gateup_q with the block terms of the 32 gate and 32 up rows parked in LDS (row stride nt + 4 floats: the 64 chain
// lanes read 16-byte pieces four banks apart) and added in block order by one lane per (matrix, row); SiLU * mul and the Q8_0 / Q8_1
// block of h as in k_gateup_q.  Dynamic LDS = 64 * (nt + 4) floats.
template <int FMT>
__global__ __launch_bounds__(1024) void k_gateup_q_ord(Planes wg, Planes wu, typename ActOf<FMT>::type act,
                                                       const unsigned short* __restrict__ exp_tab, signed char* __restrict__ q,
                                                       unsigned short* __restrict__ d, void* __restrict__ isum, int nb) {
  constexpr bool Q81 = FMT == CRABML_HIP_Q4_1;
  extern __shared__ __attribute__((aligned(16))) float ord_terms[];
  const int lane = threadIdx.x & 63, wave = wave_in_wg();
  const int blk = blockIdx.x;
  const int row = blk * 32 + wave * 2;
  const int nt = ((nb + 3) & ~3) + 4;
  // table rows: gate rows 0..31, then up rows 0..31 (the last workgroup is whole: hidden is a multiple of 32)
  rows_terms<FMT, 2>(wg.q, wg.d, act, row, 0x7fffffff, nb, lane, ord_terms + (size_t)(wave * 2) * nt, nt);
  rows_terms<FMT, 2>(wu.q, wu.d, act, row, 0x7fffffff, nb, lane, ord_terms + (size_t)(32 + wave * 2) * nt, nt);
  __syncthreads();
  if (wave == 0) {
    const float s = ordered_sum(ord_terms + (size_t)lane * nt, nb);  // lane < 32: gate row `lane`; else up row `lane - 32`
    const float u = __shfl(s, (lane & 31) + 32, 64);
    const float h = lane < 32 ? silu_mul(s, u, exp_tab) : 0.0f;
    const QLane o = quant_lane32<Q81>(h, lane < 32);  // (whole wave: the upper half quantizes zeros and stores nothing)
    if (lane < 32) {
      q[blk * 32 + lane] = o.q;
      if (lane == 0) {
        d[blk] = o.d;
        store_qaux<Q81>(isum, blk, o.aux);
      }
    }
  }
}
__global__ __launch_bounds__(256) void k_gateup_epi(const float* __restrict__ g, const float* __restrict__ u,
                                                    const unsigned short* __restrict__ exp_tab, float* __restrict__ h, int m) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < m) h[i] = silu_mul(g[i], u[i], exp_tab);
}

// batched prefill, Q8_0 / Q8_1 rhs: h = silu(g) * u quantized straight into the rows' planes (one 32-lane half-wave per block:
// quant_lane32 = the quantizer launch's arithmetic) -- the (rows, hidden) f32 h never goes to memory and back
template <bool Q81>
__global__ __launch_bounds__(256) void k_gateup_epi_quant(const float* __restrict__ g, const float* __restrict__ u,
                                                          const unsigned short* __restrict__ exp_tab, int hidden, char* __restrict__ planes,
                                                          size_t row_stride, size_t off_d, size_t off_aux) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;  // hidden % 32 == 0: half-waves are all-live or all-dead
  const size_t r = blockIdx.y;
  const bool live = i < hidden;
  const float h = live ? silu_mul(g[r * hidden + i], u[r * hidden + i], exp_tab) : 0.0f;
  const QLane o = quant_lane32<Q81>(h, live);
  if (!live) return;
  char* p = planes + r * row_stride;
  ((signed char*)p)[i] = o.q;
  if ((threadIdx.x & 31) == 0) {
    ((unsigned short*)(p + off_d))[i >> 5] = o.d;
    store_qaux<Q81>((void*)(p + off_aux), i >> 5, o.aux);
  }
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
// base: global index of logits[0] (a vocabulary shard of a tensor-parallel classifier; 0 otherwise)
__global__ __launch_bounds__(256) void k_argmax_partial(const float* __restrict__ logits, int n, float* __restrict__ pv,
                                                        int* __restrict__ pi, int base) {
  __shared__ float sv[4];
  __shared__ int si[4];
  const int per = (n + gridDim.x - 1) / gridDim.x;
  const int lo = blockIdx.x * per, hi = min(n, lo + per);
  float bv = -INFINITY;
  int bi = -1;
  for (int i = lo + threadIdx.x; i < hi; i += blockDim.x) argmax_combine(bv, bi, logits[i], base + i);
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
// best: nullable; {max as f32 bits, index} of THIS context's logits (a tensor-parallel rank's shard: the single-device
// simulation combines the ranks' pairs afterwards, k_sim_argmax_combine)
__global__ __launch_bounds__(64) void k_argmax_step(const float* __restrict__ pv, const int* __restrict__ pi, int nparts,
                                                    int* __restrict__ token_d, int* __restrict__ pos_d,
                                                    int* __restrict__ step_d, unsigned* __restrict__ out_tokens, int out_cap,
                                                    int* __restrict__ serial_d, int* __restrict__ best) {
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
    if (best != nullptr) {
      best[0] = __builtin_bit_cast(int, bv);
      best[1] = bi;
    }
    *token_d = bi;
    int st = *step_d;
    if (st < out_cap) out_tokens[st] = (unsigned)bi;
    *step_d = st + 1;
    *pos_d = *pos_d + 1;
    *serial_d = *serial_d + 1;
  }
}

// the same step for a rank of a P2P group whose classifier is split by vocabulary: the rank's {max, index} pair goes to every
// peer as two granules (slots 4 / 5 of the inbox, alternating by step: the wo / ffn_down collectives own slots 0 / 1, the host
// all-reduce 2 / 3), the n pairs are combined in rank order -- shards ascend with the rank, so "the later index wins a tie"
// (sampler.rs:109-116) holds across shards as inside one -- and every rank advances with the same token.
__global__ __launch_bounds__(64) void k_argmax_step_tp(const float* __restrict__ pv, const int* __restrict__ pi, int nparts,
                                                       int* __restrict__ token_d, int* __restrict__ pos_d, int* __restrict__ step_d,
                                                       unsigned* __restrict__ out_tokens, int out_cap, int* __restrict__ serial_d, TpP2P t,
                                                       int nseg) {
  const int lane = threadIdx.x;
  float bv = -INFINITY;
  int bi = -1;
  for (int i = lane; i < nparts; i += 64) argmax_combine(bv, bi, pv[i], pi[i]);
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    float ov = __shfl_xor(bv, o, 64);
    int oi = __shfl_xor(bi, o, 64);
    argmax_combine(bv, bi, ov, oi);
  }
  const unsigned serial = (unsigned)(*serial_d);
  const unsigned epoch = serial * (unsigned)nseg + (unsigned)nseg + t.salt;  // segment nseg - 1, + 1
  const int slot = 4 + (int)(serial & 1u);
  const unsigned base = t.cap - 2;  // the last two granules of a row
  if (lane < t.n && lane != t.me) {
#pragma unroll
    for (int p = 0; p < 8; p++)
      if (p == lane) {
        tp_put(tp_row(t, p, slot, t.me) + base, bv, epoch);
        tp_put(tp_row(t, p, slot, t.me) + base + 1, __builtin_bit_cast(float, bi), epoch);
      }
  }
  unsigned long long* mine = nullptr;
#pragma unroll
  for (int p = 0; p < 8; p++)
    if (p == t.me) mine = t.peer[p];
  float rv = bv;
  int ri = bi;
  if (lane < t.n && lane != t.me) {
    rv = tp_get(t, mine + ((size_t)slot * t.n + lane) * t.cap + base, epoch);
    ri = __builtin_bit_cast(int, tp_get(t, mine + ((size_t)slot * t.n + lane) * t.cap + base + 1, epoch));
  }
  float cv = -INFINITY;
  int ci = -1;
  for (int s = 0; s < t.n; s++) argmax_combine(cv, ci, rl_f(rv, s), __builtin_amdgcn_readlane(ri, s));
  if (lane == 0) {
    *token_d = ci;
    int st = *step_d;
    if (st < out_cap) out_tokens[st] = (unsigned)ci;
    *step_d = st + 1;
    *pos_d = *pos_d + 1;
    *serial_d = *serial_d + 1;
  }
}
// single-device simulation: the ranks' {max, index} pairs (k_argmax_step's `best`) combined in rank order; every rank's token
// word and the last entry of its token list are overwritten with the group's arg-max
struct SimBest {
  const int* best[8];
  int* token[8];
  int* step[8];
  unsigned* out_tokens[8];
};
__global__ __launch_bounds__(64) void k_sim_argmax_combine(SimBest b, int n, int out_cap) {
  if (threadIdx.x != 0) return;
  float cv = -INFINITY;
  int ci = -1;
#pragma unroll
  for (int r = 0; r < 8; r++)
    if (r < n) argmax_combine(cv, ci, __builtin_bit_cast(float, b.best[r][0]), b.best[r][1]);
#pragma unroll
  for (int r = 0; r < 8; r++)
    if (r < n) {
      *b.token[r] = ci;
      const int st = *b.step[r] - 1;  // k_argmax_step has advanced it
      if (st >= 0 && st < out_cap) b.out_tokens[r][st] = (unsigned)ci;
    }
}

}  // namespace crabml_hip

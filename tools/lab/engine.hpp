// engine.hpp -- the FFN side of a decode layer as ONE persistent launch ("engine"): wo GEMV + residual + RMSNorm + quantize ->
// gate/up GEMV + SiLU * mul + quantize -> ffn_down GEMV + residual + the next RMSNorm + quantize.
// Part of the fused decode step (fused.hip includes it after the three fused_*.hpp files).
//
// STATUS (round 3, measured on MI355X): correct -- bit-identical to the three launches it replaces at every tested shape -- and
// SLOWER: 37-39 us per launch against 5.2 + 12.8 + 9.5 = 27.5 us (600-616 vs 747 tok/s on the Llama-3-8B shape).  It is therefore
// opt-in (CRABML_HIP_LLAMA_ENGINE), kept with its tests and its profiling stamps as the measured answer to "what does the
// persistent layer of MI355X_MICROARCH.md buy on THIS path": an in-launch all-to-all edge costs ~3 us (five of them here), a
// kernel boundary ~0.6 us of gap plus a ramp, and the run-ahead is capped at the 33 MB of LDS.  DESIGN.md section 4, "Round 3".
//
// Why it was built: as three launches (k_gemv_res_nq, k_gateup_q, k_gemv_res_nq) every GEMV pays a dependent-kernel boundary and
// the ramp of its weight stream (T ~ bytes / 6.2 TB/s + 2.6 us), and the two in-launch norm gathers run with
// HBM idle.  Here one workgroup per CU stays resident for the three GEMVs and the weights never stop streaming:
//   * wave 0 is the LOADER: it copies the CU's weight stream (a contiguous, CU-major re-layout of wo | gate/up | down made once
//     at create: k_eng_pack) into an LDS ring of D slots with LDS-DMA (global_load_lds_dwordx4 ... nt, 1 KiB per instruction),
//     up to D slots (129 KiB per CU, 33 MB per chip = ~5 us of stream) ahead of the consumers -- across the two all-to-all
//     edges of the FFN, which is what a kernel boundary cannot do (MI355X_MICROARCH.md "prefetch-credit", "ldsdma-fill");
//   * waves 1.. are CONSUMERS: a slot holds R whole weight rows (quants | f16 scales); a wave takes a slot, runs the exact
//     integer block dots against the rhs planes in LDS, and frees the slot;
//   * the first consumer wave also runs the edges (8-byte {data, epoch} granules, bounded polls):
//       wo rows + residual -> row / half-chunk-sum granules -> gather -> RMSNorm + quantize the chunk -> quant granules
//       -> every CU sweeps the 9 granules per block of the normalized vector into its LDS planes            (rhs of gate/up)
//       gate/up rows -> SiLU * mul -> the wave that completes a 32-row block quantizes it -> quant granules
//       -> every CU sweeps all of h into its LDS planes                                                     (rhs of ffn_down)
//       down rows + residual -> row / half-chunk-sum granules -> gather -> RMSNorm + quantize -> the planes the next
//       layer's q/k/v launch reads.
// Arithmetic and summation orders are those of k_gemv_res_nq<FMT, 2> / k_gateup_q (lane l owns blocks l, l + 64, ... of a row
// in ascending order, wave_sum_f32 tree, quant_lane32, the 16 + 16 chunk sums, 64 chunk sums per DPP round): the engine step is
// BIT-IDENTICAL to the 5-launch step (tests/test_hip_engine.py), and therefore carries the same parity evidence against the
// oracle.  Reference ops folded in: matmul_vec (matmul_vec.rs:26-78, buf_q4_0.rs:215-253), add_inplace (arithmetic.rs:27-33),
// rms_norm_inplace + mul_inplace (rms_norm.rs:33-46, arithmetic.rs:57-66), silu_inplace (silu.rs:6-13), the per-call
// activation quantizer (buf_q8_0.rs:87-134); op order llama2.rs:600, 266, 605-638.
//
// Loader / compiler notes (ROCm 7.2): the DMA is issued from inline asm -- with the builtin the compiler drains vmcnt(0) in front
// of every LDS flag access of the loader wave -- and M0 is saved / restored inside the statement (cdna_hip_programming.md 5.7).
// Every vmcnt wait of the loader is therefore written by hand; the consumer waves never issue DMA, so their waits are the
// compiler's.  No workgroup barrier after the prologue: the waves meet through LDS sequence words only.
#pragma once
#include "fused_ffn.hpp"

namespace crabml_hip {

constexpr int ENG_SLOT = 18432;  // ring slot stride: 18 KiB = 8 rows of a 4096-column Q4_0 matrix (8 x 128 x 18 bytes)
constexpr int ENG_MAX_D = 8;     // ring depth (slots)
constexpr int ENG_MAX_BLK = 8;   // gate/up 32-row blocks per CU
constexpr int ENG_SPIN = 1 << 20;

// pointers inside EngArgs are declared global (address space 1): the struct is read from memory, and a pointer loaded from memory is
// a FLAT pointer to the compiler (flat_load / flat_store, which also tick lgkmcnt) unless its type says otherwise
#define ENG_G __attribute__((address_space(1)))
template <class T>
__host__ __device__ __forceinline__ T* eng_flat(ENG_G T* p) {  // for callees that take plain pointers (inlined: the access stays global)
  return (T*)p;
}
struct EngArgs {
  const ENG_G unsigned char* stream;        // this layer's weight stream (all CUs)
  const ENG_G unsigned long long* cu_off;   // [G + 1] byte offsets of the CUs' parts
  int G, D;                           // workgroups (= CUs used), ring depth
  int dim, nblk_h;                    // rows of wo / ffn_down; 32-row blocks of the hidden vector (local)
  int nb_wo, nb_gu, nb_dn;            // blocks per weight row: k / 32 of wo (local heads), gate/up (dim), ffn_down (local hidden)
  int R_wo, R_gu, R_dn;               // rows per slot
  int ni_wo, ni_gu, ni_dn;            // DMA pieces per slot
  int rpc, n_row_cus;                 // wo / ffn_down rows per CU (16 = half a norm chunk, or 32), CUs that own such rows
  int attn_bytes, attn_off_d, attn_off_aux;  // act_layout of the attention output's planes (global, copied to LDS)
  int dim_off_d, dim_off_aux;                // act_layout of the dim-sized rhs (LDS planes of gate/up's rhs)
  int hid_off_d, hid_off_aux;                // act_layout of the hidden-sized rhs (LDS planes of ffn_down's rhs)
  const ENG_G unsigned char* act_attn;
  ENG_G float* x;                           // residual stream (dim)
  const ENG_G float* wn_ffn;                // RMSNorm weights of the ffn norm / of the norm that follows ffn_down
  const ENG_G float* wn_next;
  float eps_ffn, eps_next;
  const ENG_G unsigned short* exp_tab;
  ENG_G signed char* oq;                    // the dim-sized rhs planes in global memory (read by the next q/k/v launch / the classifier)
  ENG_G unsigned short* od;
  ENG_G int* oisum;
  ENG_G unsigned long long* slots;          // dim / 16 half-chunk sums
  ENG_G unsigned long long* pair;           // dim row values
  ENG_G unsigned long long* xq_g;           // dim / 4 quant granules + dim / 32 scale granules of the normalized vector
  ENG_G unsigned long long* xs_g;
  ENG_G unsigned long long* hq_g;           // hidden / 4 + hidden / 32 of h
  ENG_G unsigned long long* hs_g;
  const ENG_G int* serial;                  // decode-step serial number (never reset)
  ENG_G int* fault;
  int nseg, seg0;                     // epochs: serial * nseg + seg0 + 1 (wo edge), + 2 (gate/up and down edges)
  int flags;                          // 1: thin the loader (one slot in flight) while this CU gathers
  int lag;                            // a slot is handed to the consumers once `lag` younger slots have been issued (1..3)
  ENG_G unsigned long long* stamps;   // profiling hook (CRABML_HIP_ENGINE_STAMPS=1; NULL otherwise): ENG_STAMPS words per workgroup,
                                      // s_memrealtime (100 MHz) at the phase boundaries + accumulated wait times (tools/engine_stamps.py)
};
constexpr int ENG_STAMPS = 64;
__device__ __forceinline__ unsigned long long eng_now() { return __builtin_amdgcn_s_memrealtime(); }
__device__ __forceinline__ void eng_stamp(const EngArgs& a, int c, int i, int lane) {
  if (a.stamps != nullptr && lane == 0) a.stamps[(size_t)c * ENG_STAMPS + i] = eng_now();
}
__device__ __forceinline__ void eng_stamp_val(const EngArgs& a, int c, int i, int lane, unsigned long long v) {
  if (a.stamps != nullptr && lane == 0) a.stamps[(size_t)c * ENG_STAMPS + i] = v;
}

// ---- LDS words shared by the waves of a workgroup -----------------------------------------------------------------------
__device__ __forceinline__ unsigned lds_ld(const unsigned* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }
__device__ __forceinline__ void lds_st(unsigned* p, unsigned v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }
__device__ __forceinline__ void lds_st_release(unsigned* p, unsigned v) { __hip_atomic_store(p, v, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP); }
// spin until *p >= v; false when the bound is hit or a wave of this workgroup has given up (every wait is bounded: a lost
// workgroup raises the fault word, it does not hang the device)
__device__ __forceinline__ bool lds_wait_ge(const unsigned* p, unsigned v, const unsigned* giveup) {
  for (int tries = 0; tries < ENG_SPIN; tries++) {
    if (lds_ld(p) >= v) {
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
      return true;
    }
    if (lds_ld(giveup)) return false;
    __builtin_amdgcn_s_sleep(1);
  }
  return false;
}

// ---- loader: LDS-DMA pieces, counted waits --------------------------------------------------------------------------------
// lane l copies 16 bytes from its own global address to lds_dst + 16 l (lds_dst wave-uniform, an LDS byte address); nt policy
__device__ __forceinline__ void eng_dma16(const ENG_G void* gsrc, unsigned lds_dst) {
  unsigned keep;
  asm volatile(
      "s_mov_b32 %0, m0\n\t"
      "s_mov_b32 m0, %2\n\t"
      "s_nop 0\n\t"
      "global_load_lds_dwordx4 %1, off nt\n\t"
      "s_mov_b32 m0, %0"
      : "=&s"(keep)
      : "v"(gsrc), "s"(lds_dst)
      : "memory");
}
// a whole slot of NI pieces as ONE statement: M0 is saved / restored once, the global address is SGPR base + a 32-bit VGPR offset
// that advances by 1 KiB per piece, the M0 write is separated from the DMA by the v_add (the one wait state an M0 write needs
// before an LDS-DMA instruction).  Three instructions per piece: the first loader spent ~120 cycles per piece in a compiled loop
// of twelve (0.8 us per 18 KiB slot = 21 GB/s per CU, issue-bound: profiles/r03_engine_stamps_v2.log).
#define ENG_P1 "s_addk_i32 m0, 0x400\n\tv_add_u32_e32 %1, 0x400, %1\n\tglobal_load_lds_dwordx4 %1, %2 nt\n\t"
#define ENG_P2 ENG_P1 ENG_P1
#define ENG_P4 ENG_P2 ENG_P2
#define ENG_P8 ENG_P4 ENG_P4
#define ENG_P16 ENG_P8 ENG_P8
#define ENG_T1 "s_addk_i32 m0, 0x400\n\tv_add_u32_e32 %1, 0x400, %1\n\tglobal_load_lds_dwordx4 %1, %2 nt\n\ts_waitcnt vmcnt(16)\n\t"
#define ENG_T2 ENG_T1 ENG_T1
#define ENG_T4 ENG_T2 ENG_T2
#define ENG_T8 ENG_T4 ENG_T4
#define ENG_T16 ENG_T8 ENG_T8
#define ENG_SLOT_ASM(BODY)                                                                                                     \
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2 nt\n\t" BODY "s_mov_b32 m0, %0" \
               : "=&s"(keep), "+v"(voff)                                                                                       \
               : "s"(base), "s"(lds_dst)                                                                                       \
               : "memory", "scc")
// NI = 18 or 16 pieces; base: wave-uniform; voff: this lane's byte offset of the slot's first piece; THIN: at most 17 pieces
// outstanding (the loader thinned while its CU gathers)
template <int NI, bool THIN>
__device__ __forceinline__ void eng_dma_slot(const ENG_G unsigned char* base, unsigned voff, unsigned lds_dst) {
  unsigned keep;
  if constexpr (NI == 18 && !THIN) ENG_SLOT_ASM(ENG_P16 ENG_P1);
  if constexpr (NI == 18 && THIN) ENG_SLOT_ASM(ENG_T16 ENG_T1);
  if constexpr (NI == 16 && !THIN) ENG_SLOT_ASM(ENG_P8 ENG_P4 ENG_P2 ENG_P1);
  if constexpr (NI == 16 && THIN) ENG_SLOT_ASM(ENG_T8 ENG_T4 ENG_T2 ENG_T1);
}
template <int N>
__device__ __forceinline__ void eng_wait_vm() {  // s_waitcnt vmcnt(N) only (gfx9 encoding: vmcnt = imm[3:0] | imm[15:14] << 4)
  __builtin_amdgcn_s_waitcnt(0x0F70 | (N & 15) | ((N >> 4) << 14));
}
// wait until at most n VMEM operations of this wave are outstanding (n wave-uniform; waiting for fewer is always safe)
__device__ __forceinline__ void eng_wait_vm_le(int n) {
#define ENG_W(N_) \
  case N_: eng_wait_vm<N_>(); break;
  switch (n < 0 ? 0 : n > 54 ? 54 : n) {
    ENG_W(0) ENG_W(1) ENG_W(2) ENG_W(3) ENG_W(4) ENG_W(5) ENG_W(6) ENG_W(7) ENG_W(8) ENG_W(9) ENG_W(10) ENG_W(11) ENG_W(12)
    ENG_W(13) ENG_W(14) ENG_W(15) ENG_W(16) ENG_W(17) ENG_W(18) ENG_W(19) ENG_W(20) ENG_W(21) ENG_W(22) ENG_W(23) ENG_W(24)
    ENG_W(25) ENG_W(26) ENG_W(27) ENG_W(28) ENG_W(29) ENG_W(30) ENG_W(31) ENG_W(32) ENG_W(33) ENG_W(34) ENG_W(35) ENG_W(36)
    ENG_W(37) ENG_W(38) ENG_W(39) ENG_W(40) ENG_W(41) ENG_W(42) ENG_W(43) ENG_W(44) ENG_W(45) ENG_W(46) ENG_W(47) ENG_W(48)
    ENG_W(49) ENG_W(50) ENG_W(51) ENG_W(52) ENG_W(53)
    default: eng_wait_vm<54>(); break;
  }
#undef ENG_W
}

// ---- consumers -------------------------------------------------------------------------------------------------------------
// the R row dots of one slot (R whole rows: quants [R][nb] x 16 bytes | scales [R][nb] f16) against the rhs planes in LDS;
// every lane returns the R sums.  Lane l owns blocks l, l + 64, ... in ascending order = rows_partial / k_gemv_res_nq.
template <int FMT, int R>
__device__ __forceinline__ void eng_slot_dots(const unsigned char* slot, int nb, const ActQ8_0& act, int lane, float (&out)[R]) {
  static_assert(FMT == CRABML_HIP_Q4_0, "the engine streams Q4_0 weights");
  using F = BlockFmt<FMT>;
  const i32x4* wq = (const i32x4*)slot;
  const unsigned short* wd = (const unsigned short*)(slot + (size_t)R * nb * 16);
  float acc[R];
#pragma unroll
  for (int r = 0; r < R; r++) acc[r] = 0.f;
  for (int u = lane; u < nb; u += 64) {
    const XUnit x = F::loadx(act, u);
    typename F::Blk b[R];
#pragma unroll
    for (int r = 0; r < R; r++) {
      b[r].q = wq[r * nb + u];
      b[r].d = wd[r * nb + u];
    }
#pragma unroll
    for (int r = 0; r < R; r++) acc[r] += F::term(b[r], x);
  }
#pragma unroll
  for (int r = 0; r < R; r++) out[r] = wave_sum_f32(acc[r]);
}

// what the waves of a workgroup share (static LDS)
struct EngShared {
  unsigned filled[ENG_MAX_D];  // filled[p] = s + 1 once slot s (s % D == p) has landed
  unsigned freed[ENG_MAX_D];   // freed[p]  = s + 1 once slot s has been read
  unsigned phase;              // rhs vectors gathered so far: 2 normalized x, 3 h
  unsigned giveup;             // a wave hit a bound
  unsigned gathering;          // the edge wave is polling granules
  unsigned cnt_wo, cnt_gu, cnt_dn;       // slots finished per op
  unsigned staged;                       // consumer waves that have copied their share of the attention output's planes
  unsigned edge1, sweeps;                // the wo edge has published this CU's chunk; sweep shares finished (NC per vector)
  unsigned cnt_blk[ENG_MAX_BLK];         // slots finished per gate/up block
  float xrow[32];                        // this CU's wo / ffn_down row dots
  __attribute__((aligned(16))) float hv[32];              // the edge wave's chunk (nq_epilogue's hv)
  __attribute__((aligned(16))) float hblk[ENG_MAX_BLK][64];  // (gate, up) row dots of the CU's gate/up blocks, interleaved
  __attribute__((aligned(4))) signed char qb[16][32];     // quants of a block on their way into granules (one row per wave)
};

struct EngCtx {  // per-wave view (registers)
  EngShared* S;
  unsigned char* ring;
  int D, NC, cw, lane, c;
};

__device__ __forceinline__ int eng_first(int base, int cw, int NC) {  // first stream slot >= base that wave cw owns (s % NC == cw)
  const int r = base % NC;
  return base + (cw - r + NC) % NC;
}
__device__ __forceinline__ void eng_bail(EngCtx& k, ENG_G int* fault) {
  if (k.lane == 0) {
    lds_st(&k.S->giveup, 1u);
    *fault = 1;
  }
}

// a wave's share of one op's slots.  OP 0: wo, 1: gate/up, 2: ffn_down.  s0 = the op's first stream slot, n = its slot count.
template <int FMT, int R, int OP>
__device__ __forceinline__ bool eng_consume(EngCtx& k, const EngArgs& a, int s0, int n, int nb, const ActQ8_0& act, unsigned epoch) {
  EngShared* S = k.S;
  const bool st = a.stamps != nullptr;
  unsigned long long tw = 0, tb = 0, t0 = 0, t1 = 0;
  int nslots = 0;
  for (int s = eng_first(s0, k.cw, k.NC); s < s0 + n; s += k.NC) {
    const int p = s % k.D;
    if (st) t0 = eng_now();
    if (!lds_wait_ge(&S->filled[p], (unsigned)s + 1u, &S->giveup)) return false;
    if (st) t1 = eng_now();
    float out[R];
    eng_slot_dots<FMT, R>(k.ring + (size_t)p * ENG_SLOT, nb, act, k.lane, out);
    if (k.lane == 0) lds_st_release(&S->freed[p], (unsigned)s + 1u);  // the slot's bytes are in registers (the sums depend on them)
    if (st) {
      tw += t1 - t0;
      tb += eng_now() - t1;
      nslots++;
    }
    const int j = s - s0;
    if constexpr (OP == 1) {
      // slot j of the CU's gate/up stream: block ordinal t, rows [jj R, jj R + R) of the block's 64 interleaved rows
      // (g0, u0, g1, u1, ...) = h rows jj R / 2 ...
      const int spb = 64 / R, t = j / spb, jj = j % spb;
      if (k.lane == 0) {
        // the raw (gate, up) dots are parked; SiLU * mul (one f16-table lookup per row: an L2 round trip) is taken once per
        // block by the finishing wave, 32 rows in parallel, not once per slot by one lane
#pragma unroll
        for (int r = 0; r < R; r++) S->hblk[t][jj * R + r] = out[r];
      }
      unsigned old = 0;
      if (k.lane == 0) old = __hip_atomic_fetch_add(&S->cnt_blk[t], 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_WORKGROUP);
      old = (unsigned)__builtin_amdgcn_readfirstlane((int)old);
      if (old == (unsigned)spb - 1u) {
        // this wave completed block t: quantize it (buf_q8_0.rs:87-134, k_gateup_q's half-wave) and publish it as
        // 8 {4 quants, epoch} granules + 1 {d | aux, epoch} granule (k_ffn's format)
        const int hb = k.c + t * a.G;
        const f32x2 gu = ((const f32x2*)S->hblk[t])[k.lane & 31];
        const QLane o = quant_lane32<false>(silu_mul(gu[0], gu[1], eng_flat(a.exp_tab)), true);
        if (k.lane < 32) S->qb[k.cw & 15][k.lane] = o.q;
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        __builtin_amdgcn_wave_barrier();
        if (k.lane < 8)
          __hip_atomic_store(a.hq_g + hb * 8 + k.lane,
                             ((unsigned long long)epoch << 32) | (unsigned long long)((const unsigned*)S->qb[k.cw & 15])[k.lane], __ATOMIC_RELAXED,
                             __HIP_MEMORY_SCOPE_AGENT);
        if (k.lane == 0)
          __hip_atomic_store(a.hs_g + hb, ((unsigned long long)epoch << 32) | (unsigned long long)((unsigned)o.d | (((unsigned)o.aux & 0xffffu) << 16)),
                             __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
      if (k.lane == 0) __hip_atomic_fetch_add(&S->cnt_gu, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
    } else {
      if (k.lane == 0) {
#pragma unroll
        for (int r = 0; r < R; r++) S->xrow[j * R + r] = out[r];
        __hip_atomic_fetch_add(OP == 0 ? &S->cnt_wo : &S->cnt_dn, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
      }
    }
  }
  if (st && k.cw < 3) {  // per op and consumer wave (the first three): time waiting for slots, time computing, slots
    const int base = 32 + (OP * 3 + k.cw) * 3;
    eng_stamp_val(a, k.c, base, k.lane, tw);
    eng_stamp_val(a, k.c, base + 1, k.lane, tb);
    eng_stamp_val(a, k.c, base + 2, k.lane, (unsigned long long)nslots);
  }
  return true;
}
template <int FMT, int OP>
__device__ __forceinline__ bool eng_consume_r(EngCtx& k, const EngArgs& a, int R, int s0, int n, int nb, const ActQ8_0& act, unsigned epoch) {
  switch (R) {
    case 8: return eng_consume<FMT, 8, OP>(k, a, s0, n, nb, act, epoch);
    case 4: return eng_consume<FMT, 4, OP>(k, a, s0, n, nb, act, epoch);
    case 2: return eng_consume<FMT, 2, OP>(k, a, s0, n, nb, act, epoch);
    default:
      if constexpr (OP == 1)
        return false;  // gate/up slots hold (gate, up) row pairs
      else
        return eng_consume<FMT, 1, OP>(k, a, s0, n, nb, act, epoch);
  }
}

// ---- edges (one wave) ----------------------------------------------------------------------------------------------------
// bounded poll of one granule; a granule that never arrives raises the fault word (and the workgroup's give-up word, so the
// remaining waits fall through quickly)
__device__ __forceinline__ unsigned long long eng_ldg(const ENG_G unsigned long long* p) {
  return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ unsigned eng_poll(const ENG_G unsigned long long* p, unsigned epoch, EngShared* S, ENG_G int* fault) {
  unsigned long long g = eng_ldg(p);
  int tries = 0;
  while ((unsigned)(g >> 32) != epoch && tries < (1 << 21) && !lds_ld(&S->giveup)) {
    __builtin_amdgcn_s_sleep(2);
    g = eng_ldg(p);
    tries++;
  }
  if ((unsigned)(g >> 32) != epoch) {
    *fault = 1;
    lds_st(&S->giveup, 1u);
  }
  return (unsigned)g;
}
// The all-to-all edges: every CU takes a whole quantized vector of n elements -- n / 4 {4 quants, epoch} granules and n / 32
// {d | aux, epoch} granules -- into its LDS planes q | d | isum.  ALL consumer waves of the workgroup sweep (they have nothing
// else to do until the vector is there), each a contiguous share of the granules with every load of a batch in flight at once:
// under the weight stream a granule round trip costs 1 - 3 us (MI355X_MICROARCH.md "handoff-1to1" L->L), so what is paid per
// edge is ROUND TRIPS, not bytes -- the first engine swept 16 granules per lane and pass from one wave: five dependent round
// trips for h (35 KB) where one suffices.  A batch is re-read until every granule carries the epoch (Guideline 16 R2).
constexpr int ENG_SWEEP_B = 24;  // granules per lane and batch (48 VGPRs)
__device__ __forceinline__ bool eng_sweep_part(const ENG_G unsigned long long* qg, const ENG_G unsigned long long* sg, int n, unsigned epoch,
                                               unsigned char* P, int off_d, int off_aux, int lane, int part, int nparts, EngShared* S,
                                               ENG_G int* fault) {
  unsigned* pq = (unsigned*)P;
  unsigned short* pd = (unsigned short*)(P + off_d);
  int* pa = (int*)(P + off_aux);
  const int nq = n / 4, count = nq + n / 32;
  const int per = ((count + nparts - 1) / nparts + 63) & ~63;
  const int lo = part * per, hi = lo + per < count ? lo + per : count;
  for (int base = lo; base < hi; base += 64 * ENG_SWEEP_B) {
    // a sentinel first: lane l spins on granule base + 64 * (ENG_SWEEP_B - 1) + l of the batch (clamped) -- the batch itself is
    // requested once those are in, when it is likely to be complete (a batch requested early comes back stale and is paid twice)
    {
      int j = base + 64 * (ENG_SWEEP_B - 1) + lane;
      if (j >= hi) j = hi - 1 - (lane % (hi - base < 64 ? hi - base : 64));
      (void)eng_poll(j < nq ? qg + j : sg + (j - nq), epoch, S, fault);
    }
    unsigned long long x[ENG_SWEEP_B];
    int spins = 0;
    for (;;) {
      bool ok = true;
#pragma unroll
      for (int i = 0; i < ENG_SWEEP_B; i++) {
        const int idx = base + i * 64 + lane;
        const int j = idx < hi ? idx : base;  // (base < hi: a valid granule)
        x[i] = eng_ldg(j < nq ? qg + j : sg + (j - nq));
      }
#pragma unroll
      for (int i = 0; i < ENG_SWEEP_B; i++) ok &= (unsigned)(x[i] >> 32) == epoch;
      if (__all(ok)) break;
      if (++spins > (1 << 19) || lds_ld(&S->giveup)) {
        if (lane == 0) {
          *fault = 1;
          lds_st(&S->giveup, 1u);
        }
        return false;
      }
      __builtin_amdgcn_s_sleep(2);
    }
#pragma unroll
    for (int i = 0; i < ENG_SWEEP_B; i++) {
      const int idx = base + i * 64 + lane;
      if (idx < hi) {
        const unsigned v = (unsigned)x[i];
        if (idx < nq) {
          pq[idx] = v;
        } else {
          pd[idx - nq] = (unsigned short)(v & 0xffffu);
          pa[idx - nq] = (int)(short)(v >> 16);  // |sum of 32 quants| <= 4064 fits 16 bits
        }
      }
    }
  }
  return true;
}

// The tail of wo / ffn_down for this CU's rows (nq_epilogue<FMT, SPLIT> restated for one wave per CU): xrow[] + residual -> x,
// row and half-chunk-sum granules, the one hop, RMSNorm + quantize of the chunk.  rpc = 16: two CUs share a 32-row chunk
// (SPLIT = 2), rpc = 32: the CU owns it.  LAST: the chunk's planes go to global memory (the next launch's rhs); otherwise the
// chunk's owner (part 0) publishes them as granules.  Returns the new residual of row (c rpc + lane) for lanes < rpc.
template <bool LAST>
__device__ __forceinline__ float eng_edge(EngCtx& k, const EngArgs& a, unsigned epoch, float res, float wn, float eps) {
  EngShared* S = k.S;
  const int lane = k.lane, c = k.c;
  const int split = a.rpc == 16 ? 2 : 1, ROWS = a.rpc;
  const int blk = split == 2 ? c >> 1 : c, part = split == 2 ? (c & 1) : 0;
  const int row = c * ROWS, nchunks = a.dim / 32;
  float xv = 0.f;
  if (lane < ROWS) {
    xv = S->xrow[lane] + res;  // x = matmul_out + x (llama2.rs:266 / :636)
    a.x[row + lane] = xv;
    S->hv[part * ROWS + lane] = xv;
    if (split == 2)
      __hip_atomic_store(a.pair + row + lane, ((unsigned long long)epoch << 32) | (unsigned long long)__builtin_bit_cast(unsigned, xv),
                         __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
  __builtin_amdgcn_wave_barrier();
  float cs;
  {
    float h0 = -0.0f, h1 = -0.0f;
#pragma unroll
    for (int j = 0; j < 4; j++) {
      const f32x4 t = ((const f32x4*)S->hv)[(split > 1 ? part * 4 : 0) + j];
      h0 += t[0] * t[0];
      h0 += t[1] * t[1];
      h0 += t[2] * t[2];
      h0 += t[3] * t[3];
    }
    if (split == 1) {
#pragma unroll
      for (int j = 4; j < 8; j++) {
        const f32x4 t = ((const f32x4*)S->hv)[j];
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
    __hip_atomic_store(a.slots + c, ((unsigned long long)epoch << 32) | (unsigned long long)__builtin_bit_cast(unsigned, cs), __ATOMIC_RELAXED,
                       __HIP_MEMORY_SCOPE_AGENT);
  __builtin_amdgcn_s_waitcnt(0x0F70);  // vmcnt(0): the granules are on their way before the polls queue up behind them
  if (a.flags & 3) lds_st(&S->gathering, 1u);
  const int l32 = lane & 31;
  const bool own = l32 >= part * ROWS && l32 < (part + 1) * ROWS;
  // the hop, polled in nq_epilogue's order: the partner's rows first, then every workgroup's sum in chunk order -- the wait for
  // the first straggler covers the arrival of the rest (requesting everything at once was measured slower: most granules come
  // back stale and are fetched twice; profiles/r03_engine_stamps_v3_nc7.log vs v2)
  float v = 0.0f;
  if (split > 1) {
    if (lane < 32) v = own ? S->hv[l32] : __builtin_bit_cast(float, eng_poll(a.pair + blk * 32 + l32, epoch, S, a.fault));
  } else {
    v = S->hv[l32];
  }
  float sum = 0.0f;
  for (int base = 0; base < nchunks; base += 64) {
    const int ch = base + lane;
    float cv;
    if (split > 1) {
      const float h0 = ch < nchunks ? __builtin_bit_cast(float, eng_poll(a.slots + 2 * ch, epoch, S, a.fault)) : 0.0f;
      const float h1 = ch < nchunks ? __builtin_bit_cast(float, eng_poll(a.slots + 2 * ch + 1, epoch, S, a.fault)) : 0.0f;
      cv = h0 + h1;  // chunk = its two halves (lanes past the grid add +0.0)
    } else {
      cv = ch < nchunks ? __builtin_bit_cast(float, eng_poll(a.slots + ch, epoch, S, a.fault)) : 0.0f;
    }
    sum += wave_sum_f32(cv);
  }
  const float rms = sqrtf(sum / (float)(nchunks * 32) + eps);
  const float xn = (v / rms) * wn;
  const QLane o = quant_lane32<false>(xn, true);
  if constexpr (LAST) {
    if (lane < 32 && own) {
      a.oq[blk * 32 + lane] = o.q;
      if (lane == 0) {
        a.od[blk] = o.d;
        a.oisum[blk] = o.aux;
      }
    }
  } else if (part == 0) {
    if (lane < 32) S->qb[k.cw & 15][lane] = o.q;
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __builtin_amdgcn_wave_barrier();
    if (lane < 8)
      __hip_atomic_store(a.xq_g + blk * 8 + lane, ((unsigned long long)epoch << 32) | (unsigned long long)((const unsigned*)S->qb[k.cw & 15])[lane],
                         __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (lane == 0)
      __hip_atomic_store(a.xs_g + blk, ((unsigned long long)epoch << 32) | (unsigned long long)((unsigned)o.d | (((unsigned)o.aux & 0xffffu) << 16)),
                         __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  return xv;
}

// ---- the kernel ------------------------------------------------------------------------------------------------------------
// grid = G workgroups (one per CU, all resident: checked at create), block = 64 x (1 + NC) threads, dynamic LDS = ring | planes
// ap: the layer's EngArgs in device memory (written once at create: nothing in it changes from step to step) -- 40 scalars
// and pointers as by-value kernel arguments cost the kernel ~100 live SGPRs and spills; through the pointer they are scalar
// loads where they are used
template <int FMT>
__global__ __launch_bounds__(1024) void k_engine(const EngArgs* __restrict__ ap) {
  const EngArgs& a = *ap;
  extern __shared__ __attribute__((aligned(16))) unsigned char eng_lds[];
  __shared__ EngShared S;
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const int NC = (int)(blockDim.x >> 6) - 1;
  const int c = (int)blockIdx.x;
  if (threadIdx.x < ENG_MAX_D) {
    S.filled[threadIdx.x] = 0;
    S.freed[threadIdx.x] = 0;
    S.cnt_blk[threadIdx.x] = 0;
  }
  if (threadIdx.x == 0) {
    S.phase = 0;
    S.giveup = 0;
    S.gathering = 0;
    S.cnt_wo = S.cnt_gu = S.cnt_dn = 0;
    S.edge1 = S.sweeps = S.staged = 0;
  }
  __syncthreads();  // the only workgroup barrier: no DMA is in flight yet
  // this CU's share: rows [c rpc, (c + 1) rpc) of wo and ffn_down, gate/up blocks c, c + G, ...
  const int n_wo = c < a.n_row_cus ? a.rpc / a.R_wo : 0;
  const int n_dn = c < a.n_row_cus ? a.rpc / a.R_dn : 0;
  const int nbc = c < a.nblk_h ? (a.nblk_h - c + a.G - 1) / a.G : 0;
  const int n_gu = nbc * (64 / a.R_gu);
  const int D = a.D;
  unsigned char* ACT = eng_lds + (size_t)D * ENG_SLOT;

  if (wave == 0) {
    // ================= loader =================
    __builtin_amdgcn_s_setprio(3);  // the loader's few instructions go first on the SIMD it shares with consumer waves
    const unsigned ring_base = (unsigned)(size_t)(__attribute__((address_space(3))) unsigned char*)eng_lds;
    // wave-uniform: an SGPR pair (readfirstlane: the offset comes through a vector load, which the compiler cannot prove uniform)
    const unsigned long long ba = (unsigned long long)(size_t)(a.stream + a.cu_off[c]);
    const ENG_G unsigned char* base = (const ENG_G unsigned char*)(size_t)(
        ((unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((int)(ba >> 32)) << 32) |
        (unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)ba));
    unsigned voff = (unsigned)lane * 16u;                      // this lane's byte offset inside the CU's stream
    const int lag = a.lag;
    const int total = n_wo + n_gu + n_dn;
    int published = 0;  // slots [0, published) carry their filled word
    int ni_prev = 0, ni_prev2 = 0;
    bool dead = false;
    const bool st = a.stamps != nullptr;
    unsigned long long t_full = 0, t_vm = 0, t0 = 0;
    eng_stamp(a, c, 16, lane);
    for (int s = 0; s < total && !dead; s++) {
      const int ni = s < n_wo ? a.ni_wo : s < n_wo + n_gu ? a.ni_gu : a.ni_dn;
      const int p = s % D;
      if (s >= D && lds_ld(&S.freed[p]) < (unsigned)(s - D) + 1u) {
        // the ring is full: let everything land and hand it over, then wait for the slot
        if (st) t0 = eng_now();
        eng_wait_vm<0>();
        for (; published < s; published++) lds_st(&S.filled[published % D], (unsigned)published + 1u);
        if (!lds_wait_ge(&S.freed[p], (unsigned)(s - D) + 1u, &S.giveup)) {
          dead = true;
          break;
        }
        if (st) t_full += eng_now() - t0;
      }
      if ((a.flags & 2) && lds_ld(&S.gathering) != 0) {
        // pause: while this CU's edge wave polls granules the loader issues nothing (every byte in flight queues ahead of the
        // granule loads in the CU's memory pipeline); what is in flight lands and is handed over meanwhile
        if (st) t0 = eng_now();
        eng_wait_vm<0>();
        for (; published < s; published++) lds_st(&S.filled[published % D], (unsigned)published + 1u);
        for (int tries = 0; tries < ENG_SPIN && lds_ld(&S.gathering) != 0 && !lds_ld(&S.giveup); tries++) __builtin_amdgcn_s_sleep(1);
        if (st) t_full += eng_now() - t0;
      }
      const bool thin = (a.flags & 1) && lds_ld(&S.gathering) != 0;
      const unsigned dst = ring_base + (unsigned)p * ENG_SLOT;
      if (ni == 18) {
        if (thin)
          eng_dma_slot<18, true>(base, voff, dst);
        else
          eng_dma_slot<18, false>(base, voff, dst);
      } else if (ni == 16) {
        if (thin)
          eng_dma_slot<16, true>(base, voff, dst);
        else
          eng_dma_slot<16, false>(base, voff, dst);
      } else {
#pragma unroll 2
        for (int i = 0; i < ni; i++) {
          eng_dma16(base + voff + (size_t)i * 1024, dst + (unsigned)i * 1024);
          if (thin) eng_wait_vm<16>();
        }
      }
      voff += (unsigned)ni * 1024u;
      // slot s - lag has landed once at most the pieces of the `lag` younger slots are outstanding (lag 1: ni(s); 2: + ni(s - 1);
      // 3: + ni(s - 2)).  The first slots (wo: the head of the critical path) are handed over with lag 1.
      {
        const int lg = s < n_wo + 1 ? 1 : lag;
        if (s >= lg && published <= s - lg) {
          if (st) t0 = eng_now();
          eng_wait_vm_le(lg == 1 ? ni : lg == 2 ? ni + ni_prev : ni + ni_prev + ni_prev2);
          if (st) t_vm += eng_now() - t0;
          for (; published <= s - lg; published++) lds_st(&S.filled[published % D], (unsigned)published + 1u);
        }
      }
      ni_prev2 = ni_prev;
      ni_prev = ni;
      if (st) {
        if (s == 0) eng_stamp(a, c, 17, lane);
        if (s == n_wo - 1) eng_stamp(a, c, 20, lane);
        if (s == n_wo + n_gu - 1) eng_stamp(a, c, 21, lane);
        if (s == total - 1) eng_stamp(a, c, 22, lane);
      }
    }
    if (!dead) {
      if (total >= 2 && published <= total - 2) {
        eng_wait_vm_le(ni_prev);  // everything but the last slot
        for (; published <= total - 2; published++) lds_st(&S.filled[published % D], (unsigned)published + 1u);
      }
      eng_wait_vm<0>();
      for (; published < total; published++) lds_st(&S.filled[published % D], (unsigned)published + 1u);
    } else {
      if (lane == 0) {
        *a.fault = 1;
        lds_st(&S.giveup, 1u);
      }
    }
    eng_wait_vm<0>();
    eng_stamp(a, c, 23, lane);
    eng_stamp_val(a, c, 24, lane, t_full);
    eng_stamp_val(a, c, 25, lane, t_vm);
    return;
  }

  // ================= consumers =================
  EngCtx k{&S, eng_lds, D, NC, wave - 1, lane, c};
  const unsigned e1 = (unsigned)(*a.serial) * (unsigned)a.nseg + (unsigned)a.seg0 + 1u, e2 = e1 + 1u;
  const bool edge_wave = k.cw == 0;
  const bool has_rows = c < a.n_row_cus;
  const int split = a.rpc == 16 ? 2 : 1;
  const int blk = split == 2 ? c >> 1 : c;
  float res = 0.f, wn1 = 0.f, wn2 = 0.f;
  if (edge_wave) {
    // stage the attention output's planes (q | d | isum, written by the attention launch) in LDS; the residual rows and
    // the norm weights of this CU's chunk are requested now, off the critical path
    if (has_rows) {
      if (lane < a.rpc) res = a.x[c * a.rpc + lane];
      wn1 = a.wn_ffn[blk * 32 + (lane & 31)];
      wn2 = a.wn_next[blk * 32 + (lane & 31)];
    }
    eng_stamp(a, c, 0, lane);
  }
  {
    // (every consumer wave copies a share: 16-byte pieces, up to four requests in flight per lane)
    const int n16 = a.attn_bytes / 16, step = NC * 64;
    for (int i0 = k.cw * 64 + lane; i0 < n16; i0 += 4 * step) {
      i32x4 t[4];
#pragma unroll
      for (int u = 0; u < 4; u++) t[u] = ((const ENG_G i32x4*)a.act_attn)[i0 + u * step < n16 ? i0 + u * step : i0];
#pragma unroll
      for (int u = 0; u < 4; u++)
        if (i0 + u * step < n16) ((i32x4*)ACT)[i0 + u * step] = t[u];
    }
    if (lane == 0) __hip_atomic_fetch_add(&S.staged, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
  }
  // ---- wo
  {
    if (!lds_wait_ge(&S.staged, (unsigned)NC, &S.giveup)) return eng_bail(k, a.fault);
    if (edge_wave) eng_stamp(a, c, 1, lane);
    const ActQ8_0 act{(const i32x4*)ACT, (const unsigned short*)(ACT + a.attn_off_d), (const int*)(ACT + a.attn_off_aux)};
    if (!eng_consume_r<FMT, 0>(k, a, a.R_wo, 0, n_wo, a.nb_wo, act, e1)) return eng_bail(k, a.fault);
  }
  if (edge_wave) {
    if ((a.flags & 2) && lane == 0) lds_st(&S.gathering, 1u);  // pause mode: the loader stops before the hop starts
    if (has_rows) {
      if (!lds_wait_ge(&S.cnt_wo, (unsigned)n_wo, &S.giveup)) return eng_bail(k, a.fault);
      eng_stamp(a, c, 2, lane);
      res = eng_edge<false>(k, a, e1, res, wn1, a.eps_ffn);  // x2 = wo . attn + x; its norm chunk goes out as granules
      eng_stamp(a, c, 3, lane);
    } else if (a.flags & 3) {
      lds_st(&S.gathering, 1u);
    }
    if (lane == 0) lds_st_release(&S.edge1, 1u);
  }
  // every CU takes the whole normalized vector (rhs of gate/up), all consumer waves sweeping; the planes region is free once
  // this CU's wo slots have been read (edge1 is set after cnt_wo was seen)
  if (!lds_wait_ge(&S.edge1, 1u, &S.giveup)) return eng_bail(k, a.fault);
  if (!eng_sweep_part(a.xq_g, a.xs_g, a.dim, e1, ACT, a.dim_off_d, a.dim_off_aux, lane, k.cw, NC, &S, a.fault)) return;
  if (lane == 0) {
    const unsigned done = __hip_atomic_fetch_add(&S.sweeps, 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_WORKGROUP) + 1u;
    if (done == (unsigned)NC) {
      lds_st(&S.gathering, 0u);
      lds_st_release(&S.phase, 2u);
      eng_stamp(a, c, 4, lane);
    }
  }
  // ---- gate / up
  {
    if (!lds_wait_ge(&S.phase, 2u, &S.giveup)) return eng_bail(k, a.fault);
    const ActQ8_0 act{(const i32x4*)ACT, (const unsigned short*)(ACT + a.dim_off_d), (const int*)(ACT + a.dim_off_aux)};
    if (!eng_consume_r<FMT, 1>(k, a, a.R_gu, n_wo, n_gu, a.nb_gu, act, e2)) return eng_bail(k, a.fault);
  }
  // all of h (rhs of ffn_down) -- once this CU's own gate/up slots have been read (they use the planes region)
  if (!lds_wait_ge(&S.cnt_gu, (unsigned)n_gu, &S.giveup)) return eng_bail(k, a.fault);
  if (edge_wave) {
    eng_stamp(a, c, 5, lane);
    if ((a.flags & 3) && lane == 0) lds_st(&S.gathering, 1u);
  }
  if (!eng_sweep_part(a.hq_g, a.hs_g, a.nblk_h * 32, e2, ACT, a.hid_off_d, a.hid_off_aux, lane, k.cw, NC, &S, a.fault)) return;
  if (lane == 0) {
    const unsigned done = __hip_atomic_fetch_add(&S.sweeps, 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_WORKGROUP) + 1u;
    if (done == 2u * (unsigned)NC) {
      lds_st(&S.gathering, 0u);
      lds_st_release(&S.phase, 3u);
      eng_stamp(a, c, 6, lane);
    }
  }
  // ---- ffn_down
  {
    if (!lds_wait_ge(&S.phase, 3u, &S.giveup)) return eng_bail(k, a.fault);
    const ActQ8_0 act{(const i32x4*)ACT, (const unsigned short*)(ACT + a.hid_off_d), (const int*)(ACT + a.hid_off_aux)};
    if (!eng_consume_r<FMT, 2>(k, a, a.R_dn, n_wo + n_gu, n_dn, a.nb_dn, act, e2)) return eng_bail(k, a.fault);
  }
  if (edge_wave && has_rows) {
    if (!lds_wait_ge(&S.cnt_dn, (unsigned)n_dn, &S.giveup)) return eng_bail(k, a.fault);
    eng_stamp(a, c, 7, lane);
    (void)eng_edge<true>(k, a, e2, res, wn2, a.eps_next);  // x3 = down . h + x2; the next norm's planes in global memory
    if (lane == 0) lds_st(&S.gathering, 0u);
    eng_stamp(a, c, 8, lane);
  }
}

// ---- the weight stream ------------------------------------------------------------------------------------------------------
// One thread per (CU, slot of the op, row of the slot, block of the row): copies the block's 16 quant bytes and its f16 scale
// from the matrix's planes into the slot.  op 0: wo, 1: gate (w0) / up (w1) interleaved, 2: ffn_down.  Pure byte moves.
struct EngGeom {
  int G, rpc, n_row_cus, nblk_h;
  int nb[3], R[3], ni[3];
};
__device__ __forceinline__ int eng_nslots(const EngGeom& g, int op, int c) {
  if (op == 1) return c < g.nblk_h ? ((g.nblk_h - c + g.G - 1) / g.G) * (64 / g.R[1]) : 0;
  return c < g.n_row_cus ? g.rpc / g.R[op] : 0;
}
__global__ __launch_bounds__(256) void k_eng_pack(unsigned char* __restrict__ stream, const unsigned long long* __restrict__ cu_off, EngGeom g,
                                                  int op, int nslots_max, Planes w0, Planes w1) {
  const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int nb = op == 0 ? g.nb[0] : op == 1 ? g.nb[1] : g.nb[2];
  const int R = op == 0 ? g.R[0] : op == 1 ? g.R[1] : g.R[2];
  const size_t per_cu = (size_t)nslots_max * R * nb;
  if (idx >= per_cu * g.G) return;
  const int c = (int)(idx / per_cu);
  const size_t rem = idx % per_cu;
  const int u = (int)(rem % nb), r = (int)((rem / nb) % R), j = (int)(rem / ((size_t)nb * R));
  if (j >= eng_nslots(g, op, c)) return;
  size_t off = cu_off[c];
  if (op >= 1) off += (size_t)eng_nslots(g, 0, c) * g.ni[0] * 1024;
  if (op >= 2) off += (size_t)eng_nslots(g, 1, c) * g.ni[1] * 1024;
  off += (size_t)j * (op == 0 ? g.ni[0] : op == 1 ? g.ni[1] : g.ni[2]) * 1024;
  size_t srow;
  Planes w = w0;
  if (op == 1) {
    const int spb = 64 / R, t = j / spb, jj = j % spb, wr = jj * R + r;  // wr: row of the block's 64 interleaved rows
    srow = (size_t)(c + t * g.G) * 32 + (wr >> 1);
    if (wr & 1) w = w1;
  } else {
    srow = (size_t)c * g.rpc + (size_t)j * R + r;
  }
  ((i32x4*)(stream + off))[r * nb + u] = w.q[srow * nb + u];
  ((unsigned short*)(stream + off + (size_t)R * nb * 16))[r * nb + u] = w.d[srow * nb + u];
}

}  // namespace crabml_hip

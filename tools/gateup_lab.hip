// gateup_lab.hip -- design lab (round 4): a BALANCED gate/up stage for the Q4_0 decode step on MI355X (gfx950).
//
// Not part of the product.  The shipped k_gateup_q gives one 1024-thread workgroup a whole 32-row Q8_0 quant block of h:
// 448 workgroups over 256 CUs (192 CUs carry two, 64 carry one).  Question: does a grid of small workgroups (4 / 8 / 16
// rows each, an exact multiple of the CU count) that hand the block's quantizer to the LAST ARRIVER -- 8-byte
// {h, epoch} granules + one returning atomicAdd per workgroup on the block's ticket word, no polling by anyone who is
// not last -- beat it, and by how much does the tail (granule store -> ticket -> granule read -> quantize) eat the gain?
// Every variant must produce the production kernel's bytes (q | d | isum).
//
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -o /tmp/gateup_lab tools/gateup_lab.hip && /tmp/gateup_lab
#include "../crabml_amd/csrc/fused_common.hpp"

#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <random>

using namespace crabml_hip;

#define CK(x)                                                                                \
  do {                                                                                       \
    hipError_t e_ = (x);                                                                     \
    if (e_ != hipSuccess) {                                                                  \
      fprintf(stderr, "HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); \
      exit(1);                                                                               \
    }                                                                                        \
  } while (0)

__device__ __forceinline__ float silu_mul(float g, float u, const unsigned short* __restrict__ exp_tab) {
  float nexp = exp_cached_f(-g, exp_tab);
  return (g / (1.0f + nexp)) * u;
}

// ---- V0: the production kernel (fused_ffn.hpp k_gateup_q<Q4_0>), verbatim --------------------------------------------
__global__ __launch_bounds__(1024) void k_prod(Planes wg, Planes wu, ActQ8_0 act, const unsigned short* __restrict__ exp_tab,
                                               signed char* __restrict__ q, unsigned short* __restrict__ d, int* __restrict__ isum, int nb) {
  using F = BlockFmt<CRABML_HIP_Q4_0>;
  __shared__ float hv[32];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int blk = blockIdx.x;
  const int row = blk * 32 + wave * 2;
  float g0 = 0.f, g1 = 0.f, u0 = 0.f, u1 = 0.f;
  for (int u = lane; u < nb; u += 64) {
    F::Blk bg0 = F::load(wg.q, wg.d, (size_t)row, nb, u);
    F::Blk bu0 = F::load(wu.q, wu.d, (size_t)row, nb, u);
    F::Blk bg1 = F::load(wg.q, wg.d, (size_t)row + 1, nb, u);
    F::Blk bu1 = F::load(wu.q, wu.d, (size_t)row + 1, nb, u);
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
    const QLane o = quant_lane32<false>(hv[threadIdx.x], true);
    q[blk * 32 + threadIdx.x] = o.q;
    if (threadIdx.x == 0) {
      d[blk] = o.d;
      isum[blk] = o.aux;
    }
  }
}

// ---- V1: small workgroups, last arriver quantizes -----------------------------------------------------------------------
// WPB waves per workgroup, 2 rows per wave: ROWS = 2 * WPB rows of h per workgroup, ARR = 32 / ROWS workgroups per quant block.
// MODE 0: h only (f32, no quantizer): the floor.
// MODE 1: granules + ticket, ticket taken right after the granule store was ISSUED (tags make the order irrelevant).
// MODE 2: granules, vmcnt(0), then the ticket.
// XCD: the ARR workgroups of a block get indices congruent mod 8 (same XCD under the observed dispatch).
// LDSX: the activation planes are staged in LDS once per workgroup.
struct Ticket {
  unsigned long long* gran;  // one per h row
  unsigned* tick;            // one per quant block, monotonic (every launch adds ARR)
  int* fault;
  unsigned epoch;
};
template <int WPB, int MODE, bool XCD, bool LDSX>
__global__ __launch_bounds__(64 * WPB) void k_ticket(Planes wg, Planes wu, ActQ8_0 act, const unsigned short* __restrict__ exp_tab,
                                                     signed char* __restrict__ q, unsigned short* __restrict__ d, int* __restrict__ isum,
                                                     float* __restrict__ hout, int nb, Ticket t) {
  using F = BlockFmt<CRABML_HIP_Q4_0>;
  constexpr int ROWS = 2 * WPB, ARR = 32 / ROWS;
  __shared__ __attribute__((aligned(16))) float hv[32];
  extern __shared__ __attribute__((aligned(16))) unsigned char lds_planes[];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  int blk, part;
  if constexpr (XCD) {
    const int b = blockIdx.x, grp = b / (8 * ARR), in = b % (8 * ARR);
    blk = grp * 8 + (in & 7);
    part = in >> 3;
  } else {
    blk = blockIdx.x / ARR;
    part = blockIdx.x % ARR;
  }
  const int row = blk * 32 + part * ROWS + wave * 2;
  ActQ8_0 a = act;
  if constexpr (LDSX) {
    const int k = nb * 32;
    i32x4* sq = (i32x4*)lds_planes;
    unsigned short* sd = (unsigned short*)(lds_planes + k);
    int* ss = (int*)(lds_planes + k + ((nb * 2 + 15) & ~15));
    for (int i = threadIdx.x; i < k / 16; i += 64 * WPB) sq[i] = act.q[i];
    for (int i = threadIdx.x; i < nb; i += 64 * WPB) {
      sd[i] = act.d[i];
      ss[i] = act.isum[i];
    }
    __syncthreads();
    a = ActQ8_0{sq, sd, ss};
  }
  float g0 = 0.f, g1 = 0.f, u0 = 0.f, u1 = 0.f;
  for (int u = lane; u < nb; u += 64) {
    F::Blk bg0 = F::load(wg.q, wg.d, (size_t)row, nb, u);
    F::Blk bu0 = F::load(wu.q, wu.d, (size_t)row, nb, u);
    F::Blk bg1 = F::load(wg.q, wg.d, (size_t)row + 1, nb, u);
    F::Blk bu1 = F::load(wu.q, wu.d, (size_t)row + 1, nb, u);
    const XUnit x = F::loadx(a, u);
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
    const float h0 = silu_mul(g0, u0, exp_tab), h1 = silu_mul(g1, u1, exp_tab);
    if constexpr (MODE == 0) {
      hout[row] = h0;
      hout[row + 1] = h1;
    } else {
      hv[part * ROWS + wave * 2] = h0;
      hv[part * ROWS + wave * 2 + 1] = h1;
    }
  }
  if constexpr (MODE == 0) return;
  if constexpr (WPB > 1) __syncthreads();
  if (wave != 0) return;
  if constexpr (WPB == 1) {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __builtin_amdgcn_wave_barrier();
  }
  // wave 0: the workgroup's rows as ONE coalesced granule store, then the ticket
  const unsigned epoch = t.epoch;
  const int l32 = lane & 31;
  const bool own = l32 >= part * ROWS && l32 < (part + 1) * ROWS;
  const float mine = own ? hv[l32] : 0.f;
  if (lane < 32 && own)
    __hip_atomic_store(t.gran + blk * 32 + l32, ((unsigned long long)epoch << 32) | (unsigned long long)__builtin_bit_cast(unsigned, mine),
                       __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  if constexpr (MODE == 2) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  unsigned old = 0;
  if (lane == 0) old = __hip_atomic_fetch_add(t.tick + blk, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  old = (unsigned)__builtin_amdgcn_readfirstlane((int)old);
  if ((old & (unsigned)(ARR - 1)) != (unsigned)(ARR - 1)) return;  // not the last arriver of this launch
  // last arriver: the other workgroups' rows from their granules (their tickets are in, their granules at most in flight)
  float v = mine;
  if (lane < 32 && !own) {
    const unsigned long long* p = t.gran + blk * 32 + l32;
    unsigned long long g = __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    int tries = 0;
    while ((unsigned)(g >> 32) != epoch && tries < (1 << 20)) {
      __builtin_amdgcn_s_sleep(1);
      g = __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      tries++;
    }
    if ((unsigned)(g >> 32) != epoch) *t.fault = 1;
    if (tries > 0) atomicAdd(t.fault + 1, 1);  // lab statistic: granules that were not there at the first look
    v = __builtin_bit_cast(float, (unsigned)g);
  }
  const QLane o = quant_lane32<false>(v, true);
  if (lane < 32) {
    q[blk * 32 + lane] = o.q;
    if (lane == 0) {
      d[blk] = o.d;
      isum[blk] = o.aux;
    }
  }
}

// ---- V2: the production geometry (448 x 1024) with the activation planes staged in LDS and the wave's first round of weight
// blocks requested BEFORE the staging (they depend on nothing: their HBM round trip runs under the staging and its barrier) ------
template <bool PREFETCH>
__global__ __launch_bounds__(1024) void k_prod_ldsx(Planes wg, Planes wu, ActQ8_0 act, const unsigned short* __restrict__ exp_tab,
                                                    signed char* __restrict__ q, unsigned short* __restrict__ d, int* __restrict__ isum,
                                                    int nb) {
  using F = BlockFmt<CRABML_HIP_Q4_0>;
  __shared__ float hv[32];
  extern __shared__ __attribute__((aligned(16))) unsigned char lds_planes[];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int blk = blockIdx.x;
  const int row = blk * 32 + wave * 2;
  F::Blk p0, p1, p2, p3;
  if constexpr (PREFETCH) {
    p0 = F::load(wg.q, wg.d, (size_t)row, nb, lane);
    p1 = F::load(wu.q, wu.d, (size_t)row, nb, lane);
    p2 = F::load(wg.q, wg.d, (size_t)row + 1, nb, lane);
    p3 = F::load(wu.q, wu.d, (size_t)row + 1, nb, lane);
  }
  const int k = nb * 32;
  i32x4* sq = (i32x4*)lds_planes;
  unsigned short* sd = (unsigned short*)(lds_planes + k);
  int* ss = (int*)(lds_planes + k + ((nb * 2 + 15) & ~15));
  for (int i = threadIdx.x; i < k / 16; i += 1024) sq[i] = act.q[i];
  for (int i = threadIdx.x; i < nb; i += 1024) {
    sd[i] = act.d[i];
    ss[i] = act.isum[i];
  }
  __syncthreads();
  const ActQ8_0 a{sq, sd, ss};
  float g0 = 0.f, g1 = 0.f, u0 = 0.f, u1 = 0.f;
  int u = lane;
  if constexpr (PREFETCH) {
    const XUnit x = F::loadx(a, u);
    g0 += F::term(p0, x);
    u0 += F::term(p1, x);
    g1 += F::term(p2, x);
    u1 += F::term(p3, x);
    u += 64;
  }
  for (; u < nb; u += 64) {
    F::Blk bg0 = F::load(wg.q, wg.d, (size_t)row, nb, u);
    F::Blk bu0 = F::load(wu.q, wu.d, (size_t)row, nb, u);
    F::Blk bg1 = F::load(wg.q, wg.d, (size_t)row + 1, nb, u);
    F::Blk bu1 = F::load(wu.q, wu.d, (size_t)row + 1, nb, u);
    const XUnit x = F::loadx(a, u);
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
    const QLane o = quant_lane32<false>(hv[threadIdx.x], true);
    q[blk * 32 + threadIdx.x] = o.q;
    if (threadIdx.x == 0) {
      d[blk] = o.d;
      isum[blk] = o.aux;
    }
  }
}
// the production kernel with BOTH rounds of weight blocks (k = 4096: two per row and lane) requested up front
__global__ __launch_bounds__(1024) void k_prod_all_up_front(Planes wg, Planes wu, ActQ8_0 act, const unsigned short* __restrict__ exp_tab,
                                                            signed char* __restrict__ q, unsigned short* __restrict__ d,
                                                            int* __restrict__ isum, int nb) {
  using F = BlockFmt<CRABML_HIP_Q4_0>;
  __shared__ float hv[32];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int blk = blockIdx.x;
  const int row = blk * 32 + wave * 2;
  F::Blk p[2][4];
#pragma unroll
  for (int it = 0; it < 2; it++) {
    const int u = lane + 64 * it < nb ? lane + 64 * it : nb - 1;
    p[it][0] = F::load(wg.q, wg.d, (size_t)row, nb, u);
    p[it][1] = F::load(wu.q, wu.d, (size_t)row, nb, u);
    p[it][2] = F::load(wg.q, wg.d, (size_t)row + 1, nb, u);
    p[it][3] = F::load(wu.q, wu.d, (size_t)row + 1, nb, u);
  }
  float g0 = 0.f, g1 = 0.f, u0 = 0.f, u1 = 0.f;
#pragma unroll
  for (int it = 0; it < 2; it++) {
    const int u = lane + 64 * it;
    if (u < nb) {
      const XUnit x = F::loadx(act, u);
      g0 += F::term(p[it][0], x);
      u0 += F::term(p[it][1], x);
      g1 += F::term(p[it][2], x);
      u1 += F::term(p[it][3], x);
    }
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
    const QLane o = quant_lane32<false>(hv[threadIdx.x], true);
    q[blk * 32 + threadIdx.x] = o.q;
    if (threadIdx.x == 0) {
      d[blk] = o.d;
      isum[blk] = o.aux;
    }
  }
}

// ---- exp by arithmetic instead of the 65536-entry f16 table (cpu_device.rs:108-124: table[x] = f16(expf(f32(x)))) ------------------
// The table lookup is a dependent L2 round trip at the tail of gate/up (SiLU) and inside the softmax of every attention launch.
// Candidates evaluated for ALL 65536 f16 bit patterns against the host-built table: (a) f16(expf(x)), (b) f16((float)exp((double)x)),
// (c) f16(__expf(x)).  Any candidate with zero mismatches could replace the lookup bit for bit.
__global__ void k_exp_candidates(const unsigned short* __restrict__ tab, int* __restrict__ bad) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= 65536) return;
  const float x = h2f((unsigned short)i);
  const unsigned short a = f2h(expf(x)), b = f2h((float)exp((double)x)), c = f2h(__expf(x));
  if (a != tab[i]) atomicAdd(bad + 0, 1);
  if (b != tab[i]) atomicAdd(bad + 1, 1);
  if (c != tab[i]) atomicAdd(bad + 2, 1);
  // restricted to what the decode step looks up: softmax arguments x <= 0 and SiLU arguments (any finite x)
  if (x <= 0.f && a != tab[i]) atomicAdd(bad + 3, 1);
  if (x <= 0.f && b != tab[i]) atomicAdd(bad + 4, 1);
  // harness check: a candidate that is off by 5e-4 must show up
  if (f2h(expf(x) * 1.0005f) != tab[i]) atomicAdd(bad + 5, 1);
  // ... and one 2 ulp (f32) off: what a merely "fast" exp would look like
  if (f2h(__builtin_bit_cast(float, __builtin_bit_cast(unsigned, expf(x)) + 2u)) != tab[i]) atomicAdd(bad + 6, 1);
}

// compare two (q | d | isum) triples, count mismatching blocks
__global__ void k_cmp(const signed char* q0, const unsigned short* d0, const int* s0, const signed char* q1, const unsigned short* d1,
                      const int* s1, int nblk, int* bad) {
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= nblk) return;
  bool ok = d0[b] == d1[b] && s0[b] == s1[b];
  for (int i = 0; i < 32; i++) ok = ok && q0[b * 32 + i] == q1[b * 32 + i];
  if (!ok) atomicAdd(bad, 1);
}

int main(int argc, char** argv) {
  const int dim = 4096, hidden = argc > 1 ? atoi(argv[1]) : 14336;
  const int nb = dim / 32, nblk = hidden / 32;
  hipDeviceProp_t prop;
  CK(hipGetDeviceProperties(&prop, 0));
  printf("# device %s CUs=%d  gate/up %d x %d Q4_0 (x2), %d quant blocks\n", prop.gcnArchName, prop.multiProcessorCount, hidden, dim, nblk);
  hipStream_t st;
  CK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
  // one matrix = hidden * nb blocks: 16 B quants + 2 B scale; a copy = gate | up; copies rotate so that no launch finds its
  // weights in the Infinity Cache
  const size_t qbytes = (size_t)hidden * nb * 16, dbytes = (size_t)hidden * nb * 2;
  const size_t mat = align_up(qbytes, 4096) + align_up(dbytes, 4096), copy = 2 * mat;
  const int ncopies = 10;
  char* pool;
  CK(hipMalloc(&pool, copy * ncopies));
  {
    std::vector<unsigned char> h(copy);
    std::mt19937 rng(7);
    for (size_t i = 0; i < copy; i += 4) {
      unsigned r = rng();
      memcpy(&h[i], &r, 4);
    }
    // scales: small positive / negative f16 values
    for (int m2 = 0; m2 < 2; m2++) {
      unsigned short* dd = (unsigned short*)(h.data() + m2 * mat + align_up(qbytes, 4096));
      for (size_t i = 0; i < (size_t)hidden * nb; i++) {
        _Float16 f = (_Float16)(((int)(rng() % 2001) - 1000) * 1e-5f);
        memcpy(&dd[i], &f, 2);
      }
    }
    for (int c = 0; c < ncopies; c++) CK(hipMemcpy(pool + (size_t)c * copy, h.data(), copy, hipMemcpyHostToDevice));
  }
  auto planes = [&](int c, int m2) {
    char* b = pool + (size_t)c * copy + (size_t)m2 * mat;
    return Planes{(const i32x4*)b, (const unsigned short*)(b + align_up(qbytes, 4096))};
  };
  // activation planes (Q8_0 of a random vector): q | d | isum
  const size_t off_d = dim, off_s = dim + align_up((size_t)nb * 2, 16);
  char* actp;
  CK(hipMalloc(&actp, off_s + nb * 4));
  {
    std::vector<unsigned char> h(off_s + nb * 4);
    std::mt19937 rng(11);
    for (int b = 0; b < nb; b++) {
      int s = 0;
      for (int i = 0; i < 32; i++) {
        int v = (int)(rng() % 255) - 127;
        h[b * 32 + i] = (unsigned char)(signed char)v;
        s += v;
      }
      _Float16 f = (_Float16)(0.002f + (rng() % 100) * 1e-5f);
      memcpy(&h[off_d + b * 2], &f, 2);
      memcpy(&h[off_s + b * 4], &s, 4);
    }
    CK(hipMemcpy(actp, h.data(), h.size(), hipMemcpyHostToDevice));
  }
  const ActQ8_0 act{(const i32x4*)actp, (const unsigned short*)(actp + off_d), (const int*)(actp + off_s)};
  // exp table (cpu_device.rs:108-124): exp of every f16 bit pattern, rounded to f16
  unsigned short* exp_tab;
  CK(hipMalloc(&exp_tab, 65536 * 2));
  {
    std::vector<unsigned short> h(65536);
    for (int i = 0; i < 65536; i++) {
      unsigned short bits = (unsigned short)i;
      _Float16 x;
      memcpy(&x, &bits, 2);
      _Float16 e = (_Float16)std::exp((float)x);
      memcpy(&h[i], &e, 2);
    }
    CK(hipMemcpy(exp_tab, h.data(), 65536 * 2, hipMemcpyHostToDevice));
  }
  signed char *q0, *q1;
  unsigned short *d0, *d1;
  int *s0, *s1, *bad, *fault;
  float* hout;
  unsigned long long* gran;
  unsigned* tick;
  CK(hipMalloc(&q0, hidden));
  CK(hipMalloc(&q1, hidden));
  CK(hipMalloc(&d0, nblk * 2));
  CK(hipMalloc(&d1, nblk * 2));
  CK(hipMalloc(&s0, nblk * 4));
  CK(hipMalloc(&s1, nblk * 4));
  CK(hipMalloc(&bad, 4));
  CK(hipMalloc(&fault, 8));
  CK(hipMalloc(&hout, hidden * 4));
  CK(hipMalloc(&gran, (size_t)hidden * 8));
  CK(hipMalloc(&tick, nblk * 4));
  CK(hipMemset(gran, 0, (size_t)hidden * 8));
  CK(hipMemset(tick, 0, nblk * 4));
  CK(hipMemset(fault, 0, 8));
  k_prod<<<nblk, 1024, 0, st>>>(planes(0, 0), planes(0, 1), act, exp_tab, q0, d0, s0, nb);
  CK(hipStreamSynchronize(st));
  const double MB = (2.0 * hidden * nb * 18 + 4.0 * dim + 4.0 * 2 * hidden) / 1e6;
  unsigned epoch = 1;

  auto bench = [&](const char* label, auto launch, bool check) {
    // correctness under rotation: 64 launches, outputs cleared before each, compared after each
    int nbad = 0;
    if (check) {
      for (int i = 0; i < 64; i++) {
        CK(hipMemsetAsync(q1, 0x55, hidden, st));
        CK(hipMemsetAsync(d1, 0x55, nblk * 2, st));
        CK(hipMemsetAsync(s1, 0x55, nblk * 4, st));
        CK(hipMemsetAsync(bad, 0, 4, st));
        launch(i % ncopies);
        k_cmp<<<(nblk + 255) / 256, 256, 0, st>>>(q0, d0, s0, q1, d1, s1, nblk, bad);
        int hb = 0;
        CK(hipMemcpyAsync(&hb, bad, 4, hipMemcpyDeviceToHost, st));
        CK(hipStreamSynchronize(st));
        nbad += hb;
      }
    } else {
      launch(0);
      CK(hipStreamSynchronize(st));
    }
    int f0[2] = {0, 0};
    CK(hipMemcpy(f0, fault, 8, hipMemcpyDeviceToHost));
    const int N = 240;
    for (int i = 0; i < 16; i++) launch(i % ncopies);
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    double best = 1e30, worst = 0;
    for (int rep = 0; rep < 5; rep++) {
      CK(hipEventRecord(e0, st));
      for (int i = 0; i < N; i++) launch((i + 1) % ncopies);
      CK(hipEventRecord(e1, st));
      CK(hipEventSynchronize(e1));
      float ms;
      CK(hipEventElapsedTime(&ms, e0, e1));
      best = fmin(best, ms * 1e3 / N);
      worst = fmax(worst, ms * 1e3 / N);
    }
    int f1[2] = {0, 0};
    CK(hipMemcpy(f1, fault, 8, hipMemcpyDeviceToHost));
    printf("%-58s %6.2f us [worst of 5: %6.2f]  %6.1f GB/s = %.3f of 8 TB/s  %s  fault=%d late-granules=%d (of %d launches)\n", label, best, worst,
           MB / best * 1e3, MB / best * 1e3 / 8000.0, check ? (nbad == 0 ? "bytes==prod" : "BYTES DIFFER") : "-", f1[0], f1[1] - f0[1], 5 * N + 16);
    fflush(stdout);
  };
  bench("V0 prod 448 x 1024 (one workgroup per quant block)", [&](int c) {
    k_prod<<<nblk, 1024, 0, st>>>(planes(c, 0), planes(c, 1), act, exp_tab, q1, d1, s1, nb);
  }, true);
  const size_t ldsx = off_s + nb * 4;
  bench("V2 prod 448 x 1024 + LDS activations (staged first)", [&](int c) {
    k_prod_ldsx<false><<<nblk, 1024, ldsx, st>>>(planes(c, 0), planes(c, 1), act, exp_tab, q1, d1, s1, nb);
  }, true);
  bench("V2 prod 448 x 1024 + LDS activations, weights requested first", [&](int c) {
    k_prod_ldsx<true><<<nblk, 1024, ldsx, st>>>(planes(c, 0), planes(c, 1), act, exp_tab, q1, d1, s1, nb);
  }, true);
  bench("V3 prod 448 x 1024, both rounds of weights up front", [&](int c) {
    k_prod_all_up_front<<<nblk, 1024, 0, st>>>(planes(c, 0), planes(c, 1), act, exp_tab, q1, d1, s1, nb);
  }, true);
  {
    int* badc;
    CK(hipMalloc(&badc, 7 * 4));
    CK(hipMemset(badc, 0, 7 * 4));
    k_exp_candidates<<<256, 256, 0, st>>>(exp_tab, badc);
    int hb[7];
    CK(hipMemcpyAsync(hb, badc, 7 * 4, hipMemcpyDeviceToHost, st));
    CK(hipStreamSynchronize(st));
    printf("# exp candidates vs the host table over all 65536 f16 inputs: f16(expf) %d mismatches, f16(exp double) %d, f16(__expf) %d; "
           "x <= 0 only: expf %d, double %d; harness check: expf * 1.0005 %d mismatches, expf + 2 ulp %d\n", hb[0], hb[1], hb[2], hb[3], hb[4], hb[5], hb[6]);
  }
#define RUN(WPB, MODE, XCD, LDSX, label)                                                                                              \
  bench(label, [&](int c) {                                                                                                            \
    Ticket t{gran, tick, fault, epoch++};                                                                                              \
    k_ticket<WPB, MODE, XCD, LDSX><<<hidden / (2 * WPB), 64 * WPB, LDSX ? ldsx : 0, st>>>(planes(c, 0), planes(c, 1), act, exp_tab, q1, d1, s1, \
                                                                                         hout, nb, t);                                  \
  }, MODE != 0)
  RUN(4, 0, false, false, "floor: h only, 1792 x 256");
  RUN(2, 0, false, false, "floor: h only, 3584 x 128");
  RUN(8, 0, false, false, "floor: h only,  896 x 512");
  RUN(4, 1, false, false, "ticket 1792 x 256 (4 arrivals), no wait");
  RUN(4, 2, false, false, "ticket 1792 x 256 (4 arrivals), vmcnt(0) before ticket");
  RUN(4, 1, true, false, "ticket 1792 x 256, same-XCD parts");
  RUN(2, 1, false, false, "ticket 3584 x 128 (8 arrivals), no wait");
  RUN(2, 1, true, false, "ticket 3584 x 128, same-XCD parts");
  RUN(8, 1, false, false, "ticket  896 x 512 (2 arrivals), no wait");
  RUN(1, 1, false, false, "ticket 7168 x 64 (16 arrivals), no wait");
  RUN(4, 0, false, true, "floor + LDS activations, 1792 x 256");
  RUN(4, 1, false, true, "ticket 1792 x 256 + LDS activations");
  RUN(8, 1, false, true, "ticket  896 x 512 + LDS activations");
  RUN(16, 1, false, true, "one WG per block 448 x 1024 + LDS activations");
  RUN(16, 1, false, false, "one WG per block 448 x 1024 (ticket code, 1 arrival)");
  return 0;
}

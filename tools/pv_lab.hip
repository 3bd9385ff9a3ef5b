// pv_lab.hip -- what bounds the long-context PV pass (k_attn_pv_split) per tile: the chain wave, the producers' arithmetic /
// LDS writes, or the V / P loads.  Not part of the product: a copy of the kernel with switches, timed on a synthetic f16 V
// cache of the Llama-3-8B geometry (8 kv heads x 4 q heads, head_dim 128) beside the product kernels.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -I include -o build/pv_lab tools/pv_lab.hip
#include "../crabml_amd/csrc/fused_attention.hpp"

#include <algorithm>
#include <cstdio>
#include <cstring>
#include <vector>

using namespace crabml_hip;
#define CK(x)                                                                                \
  do {                                                                                       \
    hipError_t e_ = (x);                                                                     \
    if (e_ != hipSuccess) {                                                                  \
      fprintf(stderr, "HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); \
      exit(1);                                                                               \
    }                                                                                        \
  } while (0)
namespace crabml_hip {
int set_error(crabml_hip_device*, int status, const char*, ...) { return status; }
int hip_fail(crabml_hip_device*, hipError_t, const char*, const char*, int) { return 1; }
}  // namespace crabml_hip

// the first split version (packed pairs per chain lane, four heads per workgroup), kept here for the breakdown
template <int G>
struct PvPacked {
  static constexpr int T = G <= 4 ? 256 : 128;
  static constexpr int ROW = T + 4;
  static constexpr int CHAINS = G * 16;
  static constexpr int NCW = (CHAINS + 63) / 64;
  static constexpr int THREADS = (NCW + 4) * 64;
  static constexpr size_t LDS = (size_t)2 * CHAINS * ROW * 4;
};
// MODE bits: 1 = the chain wave skips its adds (barriers only), 2 = producers skip the global loads, 4 = producers skip the
// LDS writes, 8 = producers skip the multiplies (write the V words)
template <int G, int D, int MODE>
__global__ __launch_bounds__(PvPacked<G>::THREADS) void k_pv_var(const unsigned short* __restrict__ p16, const unsigned short* __restrict__ vc,
                                                              const int* __restrict__ pos_d, float* __restrict__ out, int hd, int seq_cap) {
  typedef PvPacked<G> C;
  constexpr int T = C::T, ROW = C::ROW, CH = C::CHAINS;
  extern __shared__ __attribute__((aligned(16))) unsigned prod[];
  const int tid = threadIdx.x;
  const int nslice = hd / 32;
  const int j = blockIdx.x / nslice, sl = blockIdx.x % nslice;
  const int seq = *pos_d + 1;
  const int ntiles = (seq + T - 1) / T;
  const int nround = (ntiles + D - 1) / D * D;
  const bool chain = tid < CH;
  if (tid >= C::NCW * 64) {
    const int pt = tid - C::NCW * 64;
    const int q = pt & 3, tg = pt >> 2;
    const unsigned short* vbase = vc + (size_t)j * seq_cap * hd + sl * 32 + q * 8;
    const unsigned short* pbase = p16 + (size_t)j * G * seq_cap;
    i32x4 vr[D][4];
    unsigned long long pr[D][G];
    auto issue = [&](int tile, int s) {
      long long t0 = (long long)tile * T + 4 * tg;
      t0 = t0 + 4 <= seq_cap ? t0 : seq_cap - 4;
      if (MODE & 2) {
#pragma unroll
        for (int r = 0; r < 4; r++) vr[s][r] = i32x4{(int)t0, tile, r, s};
#pragma unroll
        for (int g = 0; g < G; g++) pr[s][g] = 0x3c003c003c003c00ull;
      } else {
#pragma unroll
        for (int r = 0; r < 4; r++) vr[s][r] = *(const i32x4*)(vbase + (size_t)(t0 + r) * hd);
#pragma unroll
        for (int g = 0; g < G; g++) pr[s][g] = *(const unsigned long long*)(pbase + (size_t)g * seq_cap + t0);
      }
    };
    auto commit = [&](int buf, int s) {
      unsigned* pb = prod + (size_t)buf * CH * ROW;
#pragma unroll
      for (int g = 0; g < G; g++) {
        unsigned pp[4];
        const unsigned lo = (unsigned)pr[s][g], hi = (unsigned)(pr[s][g] >> 32);
        pp[0] = (lo & 0xffffu) | (lo << 16);
        pp[1] = (lo >> 16) | (lo & 0xffff0000u);
        pp[2] = (hi & 0xffffu) | (hi << 16);
        pp[3] = (hi >> 16) | (hi & 0xffff0000u);
#pragma unroll
        for (int i = 0; i < 4; i++) {
          i32x4 o;
#pragma unroll
          for (int r = 0; r < 4; r++) {
            if (MODE & 8) {
              o[r] = vr[s][r][i] ^ (int)pp[r];
            } else {
              const h16x2 m = __builtin_bit_cast(h16x2, (unsigned)vr[s][r][i]) * __builtin_bit_cast(h16x2, pp[r]);
              o[r] = (int)__builtin_bit_cast(unsigned, m);
            }
          }
          if (MODE & 4) {
            if (o[0] == 0x12345678 && o[1] == 0x77777777) pb[0] = (unsigned)o[2];  // keeps the arithmetic alive
          } else {
            *(i32x4*)(pb + (size_t)(g * 16 + q * 4 + i) * ROW + 4 * tg) = o;
          }
        }
      }
    };
#pragma unroll
    for (int s = 0; s < D; s++) issue(s, s);
    commit(0, 0);
    issue(D, 0);
    __syncthreads();
    for (int base = 0; base < nround; base += D) {
#pragma unroll
      for (int u = 0; u < D; u++) {
        const int tile = base + u;
        commit((tile + 1) & 1, (u + 1) % D);
        issue(tile + 1 + D, (u + 1) % D);
        __syncthreads();
      }
    }
    return;
  }
  h16x2 c2 = {(_Float16)0.0f, (_Float16)0.0f};
  __syncthreads();
  for (int tile = 0; tile < nround; tile++) {
    if (chain && tile < ntiles && !(MODE & 1)) {
      const int nt = seq - tile * T < T ? seq - tile * T : T;
      const unsigned* row = prod + (size_t)(tile & 1) * CH * ROW + (size_t)tid * ROW;
      int t = 0;
      if (nt >= 32) {
        i32x4 a[8], b[8];
#pragma unroll
        for (int u = 0; u < 8; u++) a[u] = *(const i32x4*)(row + 4 * u);
        for (; t + 64 <= nt; t += 64) {
#pragma unroll
          for (int u = 0; u < 8; u++) b[u] = *(const i32x4*)(row + t + 32 + 4 * u);
#pragma unroll
          for (int u = 0; u < 8; u++)
#pragma unroll
            for (int r = 0; r < 4; r++) c2 = c2 + __builtin_bit_cast(h16x2, (unsigned)a[u][r]);
          {
            const int ta = t + 64 <= T - 32 ? t + 64 : T - 32;
#pragma unroll
            for (int u = 0; u < 8; u++) a[u] = *(const i32x4*)(row + ta + 4 * u);
          }
#pragma unroll
          for (int u = 0; u < 8; u++)
#pragma unroll
            for (int r = 0; r < 4; r++) c2 = c2 + __builtin_bit_cast(h16x2, (unsigned)b[u][r]);
        }
        if (t + 32 <= nt) {
#pragma unroll
          for (int u = 0; u < 8; u++)
#pragma unroll
            for (int r = 0; r < 4; r++) c2 = c2 + __builtin_bit_cast(h16x2, (unsigned)a[u][r]);
          t += 32;
        }
      }
      for (; t + 4 <= nt; t += 4) {
        const i32x4 v = *(const i32x4*)(row + t);
#pragma unroll
        for (int r = 0; r < 4; r++) c2 = c2 + __builtin_bit_cast(h16x2, (unsigned)v[r]);
      }
      for (; t < nt; t++) c2 = c2 + __builtin_bit_cast(h16x2, row[t]);
    }
    __syncthreads();
  }
  if (!chain) return;
  const int g = tid >> 4, dp = tid & 15;
  const int e0 = (j * G + g) * hd + sl * 32 + 2 * dp;
  out[e0] = (float)c2[0];
  out[e0 + 1] = (float)c2[1];
}


// v3: ONE dim per lane (v_add_f16: 7.1 cycles dependent against 10.4 for v_pk_add_f16, valu_chain_lab), chains = G x 32 dims
template <int G, int HG = G>
struct Pv3 {
  static constexpr int T = HG <= 4 ? 256 : 128;
  static constexpr int ROWB = T * 2 + 16;           // bytes per chain row (conflict-free ds_read_b128 across lanes)
  static constexpr int CHAINS = HG * 32;
  static constexpr int NCW = (CHAINS + 63) / 64;
  static constexpr int THREADS = (NCW + 4) * 64;
  static constexpr size_t LDS = (size_t)2 * CHAINS * ROWB;
};
template <int G, int MODE, int HG = G>
__global__ __launch_bounds__((Pv3<G, HG>::THREADS)) void k_pv3(const unsigned short* __restrict__ p16, const unsigned short* __restrict__ vc,
                                                       const int* __restrict__ pos_d, float* __restrict__ out, int hd, int seq_cap) {
  typedef Pv3<G, HG> C;
  constexpr int T = C::T, ROWB = C::ROWB, CH = C::CHAINS, D = 3, NSUB = G / HG;
  extern __shared__ __attribute__((aligned(16))) unsigned char prodb[];
  const int tid = threadIdx.x;
  const int nslice = hd / 32;
  const int hsub = blockIdx.x % NSUB;
  const int j = blockIdx.x / NSUB / nslice, sl = blockIdx.x / NSUB % nslice;
  const int seq = *pos_d + 1;
  const int ntiles = (seq + T - 1) / T;
  const int nround = (ntiles + D - 1) / D * D;
  const bool chain = tid < CH;
  if (tid >= C::NCW * 64) {
    const int pt = tid - C::NCW * 64;
    const int q = pt & 3, tg = pt >> 2;
    const bool live = tg < T / 4;
    if (!live) {
      for (int tile = 0; tile <= nround; tile++) __syncthreads();
      return;
    }
    const unsigned short* vbase = vc + (size_t)j * seq_cap * hd + sl * 32 + q * 8;
    const unsigned short* pbase = p16 + (size_t)(j * G + hsub * HG) * seq_cap;
    i32x4 vr[D][4];
    unsigned long long pr[D][HG];
    auto issue = [&](int tile, int s) {
      long long t0 = (long long)tile * T + 4 * tg;
      t0 = t0 + 4 <= seq_cap ? t0 : seq_cap - 4;
#pragma unroll
      for (int r = 0; r < 4; r++) vr[s][r] = *(const i32x4*)(vbase + (size_t)(t0 + r) * hd);
#pragma unroll
      for (int g = 0; g < HG; g++) pr[s][g] = *(const unsigned long long*)(pbase + (size_t)g * seq_cap + t0);
    };
    auto commit = [&](int buf, int s) {
      unsigned char* pb = prodb + (size_t)buf * CH * ROWB + 8 * tg;
#pragma unroll
      for (int g = 0; g < HG; g++) {
        unsigned pp[4];
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
          // dim 2i (low halves) and dim 2i + 1 (high halves) of the four positions
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
    for (int base = 0; base < nround; base += D) {
#pragma unroll
      for (int u = 0; u < D; u++) {
        const int tile = base + u;
        commit((tile + 1) & 1, (u + 1) % D);
        issue(tile + 1 + D, (u + 1) % D);
        __syncthreads();
      }
    }
    return;
  }
  if (MODE & 16) __builtin_amdgcn_s_setprio(3);
  _Float16 c = (_Float16)0.0f;
  __syncthreads();
#define ADD8(w)                                                   \
  _Pragma("unroll") for (int r_ = 0; r_ < 4; r_++) {             \
    const h16x2 p_ = __builtin_bit_cast(h16x2, (unsigned)(w)[r_]); \
    c = c + p_[0];                                                \
    c = c + p_[1];                                                \
  }
  for (int tile = 0; tile < nround; tile++) {
    if (chain && tile < ntiles) {
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
          for (int u = 0; u < 8; u++) ADD8(a[u])
          {
            const int ta = t + 128 <= T - 64 ? t + 128 : T - 64;
#pragma unroll
            for (int u = 0; u < 8; u++) a[u] = *(const i32x4*)(row + 2 * ta + 16 * u);
          }
#pragma unroll
          for (int u = 0; u < 8; u++) ADD8(b[u])
        }
        if (t + 64 <= nt) {
#pragma unroll
          for (int u = 0; u < 8; u++) ADD8(a[u])
          t += 64;
        }
      }
      for (; t + 8 <= nt; t += 8) {
        const i32x4 v = *(const i32x4*)(row + 2 * t);
        ADD8(v)
      }
      for (; t < nt; t++) c = c + *(const _Float16*)(row + 2 * t);
    }
    __syncthreads();
  }
#undef ADD8
  if (!chain) return;
  const int g = hsub * HG + (tid >> 5), d = tid & 31;
  out[(j * G + g) * hd + sl * 32 + d] = (float)c;
}

template <typename F>
static float time_us(hipStream_t st, int reps, F&& launch) {
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  for (int i = 0; i < 3; i++) launch(i);
  CK(hipStreamSynchronize(st));
  CK(hipEventRecord(e0, st));
  for (int i = 0; i < reps; i++) launch(i);
  CK(hipEventRecord(e1, st));
  CK(hipStreamSynchronize(st));
  float ms;
  CK(hipEventElapsedTime(&ms, e0, e1));
  return ms * 1e3f / reps;
}

int main() {
  constexpr int G = 4;
  const int n_kv = 8, hd = 128, seq_cap = 8192, n_heads = n_kv * G;
  const int L = 24;
  hipStream_t st;
  CK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
  const size_t kvb = (size_t)n_kv * seq_cap * hd * 2;
  char* vc;
  CK(hipMalloc(&vc, kvb * L));
  std::vector<unsigned short> h(kvb / 2);
  for (size_t i = 0; i < h.size(); i++) h[i] = (unsigned short)(0x3000 + (i * 2654435761u >> 20) % 0x0800);
  for (int l = 0; l < L; l++) CK(hipMemcpy(vc + l * kvb, h.data(), kvb, hipMemcpyHostToDevice));
  unsigned short* p16;
  CK(hipMalloc(&p16, (size_t)n_heads * seq_cap * 2));
  std::vector<unsigned short> hp((size_t)n_heads * seq_cap);
  for (size_t i = 0; i < hp.size(); i++) hp[i] = (unsigned short)(0x1000 + (i * 40503u >> 8) % 0x0400);  // tiny positive f16
  CK(hipMemcpy(p16, hp.data(), hp.size() * 2, hipMemcpyHostToDevice));
  float *out, *out2;
  CK(hipMalloc(&out, n_heads * hd * 4));
  CK(hipMalloc(&out2, n_heads * hd * 4));
  int* pos_d;
  CK(hipMalloc(&pos_d, 4));
  const dim3 grid(n_kv * (hd / 32));
  CK(hipFuncSetAttribute((const void*)k_attn_pv_split<G>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)PvSplit<G>::LDS));
#define SETATTR(D_, M_) CK(hipFuncSetAttribute((const void*)k_pv_var<G, D_, M_>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)PvPacked<G>::LDS))
#define RUNVAR(D_, M_, label)                                                                                                            \
  {                                                                                                                                      \
    SETATTR(D_, M_);                                                                                                                     \
    const float us = time_us(st, 48, [&](int i) {                                                                                        \
      k_pv_var<G, D_, M_><<<grid, PvPacked<G>::THREADS, PvPacked<G>::LDS, st>>>(p16, (const unsigned short*)(vc + (size_t)(i % L) * kvb), \
                                                                             pos_d, out2, hd, seq_cap);                                 \
    });                                                                                                                                  \
    printf("  %-58s %7.2f us  (%5.1f cycles/position at 2.4 GHz)\n", label, us, us * 2400.0 / seq);                                      \
  }
  for (int seq : {1024, 4096, 8000}) {
    const int pos = seq - 1;
    CK(hipMemcpy(pos_d, &pos, 4, hipMemcpyHostToDevice));
    printf("seq %d\n", seq);
    {
      const float us = time_us(st, 48, [&](int i) {
        k_attn_pv<G><<<grid, 256, 0, st>>>(p16, (const unsigned short*)(vc + (size_t)(i % L) * kvb), pos_d, out, nullptr, nullptr, nullptr, hd,
                                           seq_cap, 0, 0);
      });
      printf("  %-58s %7.2f us  (%5.1f cycles/position at 2.4 GHz)\n", "k_attn_pv (chain wave multiplies and adds)", us, us * 2400.0 / seq);
    }
    {
      const float us = time_us(st, 48, [&](int i) {
        k_attn_pv_split<G><<<dim3(grid.x * PvSplit<G>::NSUB), PvSplit<G>::THREADS, PvSplit<G>::LDS, st>>>(p16, (const unsigned short*)(vc + (size_t)(i % L) * kvb), pos_d, out2,
                                                                              nullptr, nullptr, nullptr, hd, seq_cap, 0, 0);
      });
      printf("  %-58s %7.2f us  (%5.1f cycles/position at 2.4 GHz)\n", "k_attn_pv_split (product: one column per lane, 2 heads/WG)", us, us * 2400.0 / seq);
    }
    {
      CK(hipFuncSetAttribute((const void*)k_pv3<G, 0>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)Pv3<G>::LDS));
      CK(hipFuncSetAttribute((const void*)k_pv3<G, 16>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)Pv3<G>::LDS));
      CK(hipMemsetAsync(out2, 0, n_heads * hd * 4, st));
      float us = time_us(st, 48, [&](int i) {
        k_pv3<G, 0><<<grid, Pv3<G>::THREADS, Pv3<G>::LDS, st>>>(p16, (const unsigned short*)(vc + (size_t)(i % L) * kvb), pos_d, out2, hd, seq_cap);
      });
      printf("  %-58s %7.2f us  (%5.1f cycles/position at 2.4 GHz)\n", "v3: one dim per lane, v_add_f16 chain", us, us * 2400.0 / seq);
      us = time_us(st, 48, [&](int i) {
        k_pv3<G, 16><<<grid, Pv3<G>::THREADS, Pv3<G>::LDS, st>>>(p16, (const unsigned short*)(vc + (size_t)(i % L) * kvb), pos_d, out2, hd, seq_cap);
      });
      printf("  %-58s %7.2f us  (%5.1f cycles/position at 2.4 GHz)\n", "v3 + s_setprio 3 on the chain waves", us, us * 2400.0 / seq);
      CK(hipFuncSetAttribute((const void*)k_pv3<G, 0, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)Pv3<G, 2>::LDS));
      CK(hipFuncSetAttribute((const void*)k_pv3<G, 0, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)Pv3<G, 1>::LDS));
      us = time_us(st, 48, [&](int i) {
        k_pv3<G, 0, 2><<<dim3(grid.x * 2), Pv3<G, 2>::THREADS, Pv3<G, 2>::LDS, st>>>(p16, (const unsigned short*)(vc + (size_t)(i % L) * kvb), pos_d,
                                                                                  out2, hd, seq_cap);
      });
      printf("  %-58s %7.2f us  (%5.1f cycles/position at 2.4 GHz)\n", "v4: two heads per workgroup (64 chains, 64 WGs)", us, us * 2400.0 / seq);
      us = time_us(st, 48, [&](int i) {
        k_pv3<G, 0, 1><<<dim3(grid.x * 4), Pv3<G, 1>::THREADS, Pv3<G, 1>::LDS, st>>>(p16, (const unsigned short*)(vc + (size_t)(i % L) * kvb), pos_d,
                                                                                  out2, hd, seq_cap);
      });
      printf("  %-58s %7.2f us  (%5.1f cycles/position at 2.4 GHz)\n", "v4: one head per workgroup (32 chains, 128 WGs)", us, us * 2400.0 / seq);
      CK(hipMemsetAsync(out2, 0, n_heads * hd * 4, st));
      k_pv3<G, 0, 2><<<dim3(grid.x * 2), Pv3<G, 2>::THREADS, Pv3<G, 2>::LDS, st>>>(p16, (const unsigned short*)vc, pos_d, out2, hd, seq_cap);
      k_attn_pv<G><<<grid, 256, 0, st>>>(p16, (const unsigned short*)vc, pos_d, out, nullptr, nullptr, nullptr, hd, seq_cap, 0, 0);
      CK(hipStreamSynchronize(st));
      {
        std::vector<float> a(n_heads * hd), b(n_heads * hd);
        CK(hipMemcpy(a.data(), out, a.size() * 4, hipMemcpyDeviceToHost));
        CK(hipMemcpy(b.data(), out2, b.size() * 4, hipMemcpyDeviceToHost));
        int bad = 0;
        for (size_t i = 0; i < a.size(); i++) bad += memcmp(&a[i], &b[i], 4) != 0;
        printf("  v4 (two heads) vs k_attn_pv: %d of %zu outputs differ\n", bad, a.size());
      }
      // same cache copy for the comparison
      k_attn_pv<G><<<grid, 256, 0, st>>>(p16, (const unsigned short*)vc, pos_d, out, nullptr, nullptr, nullptr, hd, seq_cap, 0, 0);
      k_pv3<G, 0><<<grid, Pv3<G>::THREADS, Pv3<G>::LDS, st>>>(p16, (const unsigned short*)vc, pos_d, out2, hd, seq_cap);
      CK(hipStreamSynchronize(st));
      std::vector<float> a(n_heads * hd), b(n_heads * hd);
      CK(hipMemcpy(a.data(), out, a.size() * 4, hipMemcpyDeviceToHost));
      CK(hipMemcpy(b.data(), out2, b.size() * 4, hipMemcpyDeviceToHost));
      int bad = 0;
      for (size_t i = 0; i < a.size(); i++) bad += memcmp(&a[i], &b[i], 4) != 0;
      printf("  v3 vs k_attn_pv: %d of %zu outputs differ (sample %g %g)\n", bad, a.size(), a[5], b[5]);
    }
    RUNVAR(3, 0, "packed pairs, four heads per workgroup, D = 3");
    RUNVAR(6, 0, "packed pairs, D = 6 tiles of loads in flight");
    RUNVAR(3, 1, "D = 3, chain wave idle");
    RUNVAR(3, 2, "D = 3, no global loads");
    RUNVAR(3, 2 | 1, "D = 3, no global loads, chain idle");
    RUNVAR(3, 2 | 4, "D = 3, no global loads, no LDS writes");
    RUNVAR(3, 2 | 4 | 1, "D = 3, no loads, no LDS writes, chain idle (barriers)");
    RUNVAR(3, 2 | 8, "D = 3, no global loads, no multiplies");
    std::vector<float> a(n_heads * hd), b(n_heads * hd);
    CK(hipMemcpy(a.data(), out, a.size() * 4, hipMemcpyDeviceToHost));
    CK(hipMemcpy(b.data(), out2, b.size() * 4, hipMemcpyDeviceToHost));
  }
  return 0;
}

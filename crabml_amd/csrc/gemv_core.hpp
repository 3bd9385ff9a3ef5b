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
  const i32x4* q;
  const float* d;
  const short* bsums;
};

// ---- per-format block access for the 32-element formats whose rhs is Q8_0 -----------------------
template <int FMT>
struct BlockFmt;

template <>
struct BlockFmt<CRABML_HIP_Q4_0> {
  struct Blk {
    i32x4 q;
    unsigned short d;
  };
  static __device__ __forceinline__ Blk load(const i32x4* wq, const unsigned short* wd, size_t idx) {
    Blk b;
    b.q = __builtin_nontemporal_load(wq + idx);
    b.d = __builtin_nontemporal_load(wd + idx);
    return b;
  }
  // buf_q4_0.rs:249: sumi as f32 * d_w * d_x
  static __device__ __forceinline__ float term(const Blk& b, i32x4 x0, i32x4 x1, float dx, int xs) {
    return ((float)dot_q4_0(b.q, x0, x1, xs) * h2f(b.d)) * dx;
  }
};

template <>
struct BlockFmt<CRABML_HIP_Q8_0> {
  struct Blk {
    i32x4 q0, q1;
    unsigned short d;
  };
  static __device__ __forceinline__ Blk load(const i32x4* wq, const unsigned short* wd, size_t idx) {
    Blk b;
    b.q0 = __builtin_nontemporal_load(wq + 2 * idx);
    b.q1 = __builtin_nontemporal_load(wq + 2 * idx + 1);
    b.d = __builtin_nontemporal_load(wd + idx);
    return b;
  }
  // buf_q8_0.rs:282
  static __device__ __forceinline__ float term(const Blk& b, i32x4 x0, i32x4 x1, float dx, int) {
    return ((float)dot_i8x32(b.q0, b.q1, x0, x1) * h2f(b.d)) * dx;
  }
};

// Per-lane partial sums of R rows of one weight matrix against one quantized activation vector: lane l
// owns blocks l, l+64, ...  The R block loads of a step are issued before any is consumed.
template <int FMT, int R>
__device__ __forceinline__ void rows_partial(const i32x4* __restrict__ wq, const unsigned short* __restrict__ wd,
                                             const ActQ8_0& act, int row0, int m, int nb, int lane, float acc[R]) {
  using F = BlockFmt<FMT>;
#pragma unroll
  for (int r = 0; r < R; r++) acc[r] = 0.f;
  for (int b = lane; b < nb; b += 64) {
    typename F::Blk blk[R];
#pragma unroll
    for (int r = 0; r < R; r++) {
      int row = row0 + r < m ? row0 + r : m - 1;
      blk[r] = F::load(wq, wd, (size_t)row * nb + b);
    }
    i32x4 x0 = act.q[2 * b], x1 = act.q[2 * b + 1];
    float dx = h2f(act.d[b]);
    int xs = act.isum[b];
#pragma unroll
    for (int r = 0; r < R; r++) acc[r] += F::term(blk[r], x0, x1, dx, xs);
  }
}

}  // namespace crabml_hip

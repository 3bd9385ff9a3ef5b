// devutil.hpp -- small device helpers shared by the kernels (gfx950: wave = 64 lanes).
#pragma once
#include "common.hpp"

namespace crabml_hip {

// half crate semantics: from_f32 = IEEE RNE (v_cvt_f16_f32 in the default round mode), to_f32 exact.
__device__ __forceinline__ float h2f(unsigned short h) {
  _Float16 x;
  __builtin_memcpy(&x, &h, 2);
  return (float)x;
}
__device__ __forceinline__ unsigned short f2h(float f) {
  // The empty asm pins the f32 value in a VGPR first.  Without it hipcc folds f16(a*b) / f16(a+b) into
  // v_fma_mixlo_f16 (a*b + 0 rounded ONCE to f16): that drops the sign of a -0.0 product and skips the
  // intermediate f32 rounding the half crate performs -- seen on MI355X as Q8_1 `s` = +0 instead of -0.
  asm("" : "+v"(f));  // (not volatile: may be scheduled freely, only the value is opaque)
  _Float16 x = (_Float16)f;
  unsigned short h;
  __builtin_memcpy(&h, &x, 2);
  return h;
}
// half: `a * b` / `a + b` on f16 = compute in f32, round once to f16
__device__ __forceinline__ unsigned short h_mul(unsigned short a, unsigned short b) { return f2h(h2f(a) * h2f(b)); }
__device__ __forceinline__ unsigned short h_add(unsigned short a, unsigned short b) { return f2h(h2f(a) + h2f(b)); }

// Native f16 arithmetic.  The half crate computes `a * b` / `a + b` in f32 and rounds once to f16; for + and * that
// double rounding is innocuous (24 >= 2 * 11 + 2 significand bits), so it equals ONE correctly rounded f16
// operation: v_mul_f16 / v_add_f16.  Used by the f16-accumulated PV chain of attention (buf_f16.rs:152-163), where
// the add is a serial dependency per cached position (1 dependent instruction instead of 3).  Compiled with
// -ffp-contract=off: the product and the sum stay two roundings.
__device__ __forceinline__ _Float16 hbits(unsigned short h) {
  _Float16 x;
  __builtin_memcpy(&x, &h, 2);
  return x;
}
__device__ __forceinline__ unsigned short hraw(_Float16 x) {
  unsigned short h;
  __builtin_memcpy(&h, &x, 2);
  return h;
}

// ---- cross-lane reductions on DPP (gfx9 data-parallel primitives) -------------------------------------
// __shfl_xor lowers to ds_bpermute_b32 (an LDS-crossbar round trip, ~100 cycles each, and a 5-6 deep
// dependent chain per reduction); measured on MI355X this made the single-workgroup norm+quantize stage
// 13.8 us.  DPP row operations are register-to-register: 4 steps reduce each 16-lane row, v_readlane
// combines the 4 rows.  All lanes must be active (callers keep whole waves converged).
// the wave's index inside its workgroup as a SCALAR (threadIdx.x >> 6 is uniform per wave, but the compiler cannot know: everything
// derived from it -- the rows a wave owns, their base addresses -- then lives in VGPRs and every address is 64-bit vector arithmetic;
// through readfirstlane the row bases become SGPR pairs and the loads take the `saddr + 32-bit lane offset` form)
__device__ __forceinline__ int wave_in_wg() { return __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)); }

template <int CTRL>
__device__ __forceinline__ int dpp_i(int v) {
  return __builtin_amdgcn_update_dpp(0, v, CTRL, 0xF, 0xF, false);
}
template <int CTRL>
__device__ __forceinline__ float dpp_f(float v) {
  return __builtin_bit_cast(float, dpp_i<CTRL>(__builtin_bit_cast(int, v)));
}
// DPP controls: quad_perm[1,0,3,2]=0xB1, quad_perm[2,3,0,1]=0x4E, row_half_mirror=0x141, row_mirror=0x140
__device__ __forceinline__ float row16_sum_f32(float v) {
  v += dpp_f<0xB1>(v);
  v += dpp_f<0x4E>(v);
  v += dpp_f<0x141>(v);
  v += dpp_f<0x140>(v);
  return v;
}
__device__ __forceinline__ float row16_max_f32(float v) {
  v = fmaxf(v, dpp_f<0xB1>(v));
  v = fmaxf(v, dpp_f<0x4E>(v));
  v = fmaxf(v, dpp_f<0x141>(v));
  v = fmaxf(v, dpp_f<0x140>(v));
  return v;
}
__device__ __forceinline__ int row16_sum_i32(int v) {
  v += dpp_i<0xB1>(v);
  v += dpp_i<0x4E>(v);
  v += dpp_i<0x141>(v);
  v += dpp_i<0x140>(v);
  return v;
}
__device__ __forceinline__ float rl_f(float v, int lane) {
  return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), lane));
}
// full-wave (64 lanes): every lane gets the result
__device__ __forceinline__ float wave_sum_f32(float v) {
  v = row16_sum_f32(v);
  return (rl_f(v, 0) + rl_f(v, 16)) + (rl_f(v, 32) + rl_f(v, 48));
}
__device__ __forceinline__ float wave_max_f32(float v) {
  v = row16_max_f32(v);
  return fmaxf(fmaxf(rl_f(v, 0), rl_f(v, 16)), fmaxf(rl_f(v, 32), rl_f(v, 48)));
}
// per 32-lane half (one Q8_0 block per half-wave): lanes 0-31 get their half's result, lanes 32-63 theirs
__device__ __forceinline__ float half_max_f32(float v) {
  v = row16_max_f32(v);
  float lo = fmaxf(rl_f(v, 0), rl_f(v, 16)), hi = fmaxf(rl_f(v, 32), rl_f(v, 48));
  return (threadIdx.x & 32) ? hi : lo;
}
__device__ __forceinline__ int half_sum_i32(int v) {
  v = row16_sum_i32(v);
  int lo = __builtin_amdgcn_readlane(v, 0) + __builtin_amdgcn_readlane(v, 16);
  int hi = __builtin_amdgcn_readlane(v, 32) + __builtin_amdgcn_readlane(v, 48);
  return (threadIdx.x & 32) ? hi : lo;
}

// Q4_K header words u0..u2 (the re-packed scales, common.hpp) -> the 24-bit field of pair p:
// scale[2p] | scale[2p+1] << 6 | min[2p] << 12 | min[2p+1] << 18
__device__ __forceinline__ unsigned q4k_pair_field(unsigned u0, unsigned u1, unsigned u2, int p) {
  const unsigned lo = p == 0 ? u0 : p == 1 ? u0 : p == 2 ? u1 : u2;
  const unsigned hi = p == 0 ? u0 : p == 1 ? u1 : p == 2 ? u2 : u2;
  const unsigned sh = p == 0 ? 0u : p == 1 ? 24u : p == 2 ? 16u : 8u;
  return (unsigned)(((((unsigned long long)hi) << 32) | lo) >> sh) & 0xffffffu;
}

// Rust `f32 as i32`: saturating, NaN -> 0
__device__ __forceinline__ int rs_f32_as_i32(float v) {
  if (v != v) return 0;
  if (v >= 2147483648.0f) return 2147483647;
  if (v <= -2147483648.0f) return (-2147483647 - 1);
  return (int)v;
}

template <bool NT, typename T>
__device__ __forceinline__ T ld(const T* p) {
  if constexpr (NT)
    return __builtin_nontemporal_load(p);
  else
    return *p;
}

// Q8_K quantizer of one 256-element super-block by one wave, 4 consecutive elements per lane
// (buf_q8_k.rs:84-131: scale = -128 / (the FIRST element of maximal |x|), round half away from zero, min(127),
// `as i8` saturation).  Returns the 4 quants packed into a dword; `quad_sum` = this lane's quad's sum of 16 quants
// (bsums entry lane/4), `d` = the block scale (0 for an all-zero block).
struct Q8KLane {
  unsigned packed;
  int quad_sum;
  float d;
};
__device__ __forceinline__ Q8KLane q8k_wave_quant(const f32x4 v, int lane) {
  // first element of maximal |x|: the lane's own first maximum (strict `>` scan), the wave maximum by DPP, then
  // the lowest lane that holds it (ballot) hands over its element
  float loc_abs = 0.f, loc_val = 0.f;
#pragma unroll
  for (int i = 0; i < 4; i++) {
    float a = fabsf(v[i]);
    if (a > loc_abs) {
      loc_abs = a;
      loc_val = v[i];
    }
  }
  const float best_abs = wave_max_f32(loc_abs);
  const unsigned long long holders = __ballot(loc_abs == best_abs);
  const int first = holders ? __builtin_ctzll(holders) : 0;
  const float best_val = rl_f(loc_val, first);
  const float scale = -128.0f / best_val;
  Q8KLane o;
  o.d = best_abs == 0.0f ? 0.0f : 1.0f / scale;
  int qi[4];
  int s = 0;
#pragma unroll
  for (int i = 0; i < 4; i++) {
    int t = 0;
    if (best_abs != 0.0f) {
      float r = roundf(scale * v[i]);  // half away from zero
      r = fminf(r, 127.0f);
      t = rs_f32_as_i32(r);
      t = t < -128 ? -128 : t;  // `as i8` saturates
    }
    qi[i] = t;
    s += t;
  }
  o.packed = ((unsigned)qi[0] & 0xffu) | (((unsigned)qi[1] & 0xffu) << 8) | (((unsigned)qi[2] & 0xffu) << 16) |
             (((unsigned)qi[3] & 0xffu) << 24);
  s += dpp_i<0xB1>(s);
  s += dpp_i<0x4E>(s);
  o.quad_sum = s;
  return o;
}

// The class-major copy of a super-block's quants (plane `qp`, common.hpp): q8k_wave_quant leaves lane L the four consecutive
// elements 4 L .. 4 L + 3; element 4 j + k of 32-group c = L / 8 (j = L % 8) goes to byte 4 ((4 j + k) % 8) + (4 j + k) / 8 =
// 16 (j & 1) + 4 k + j / 2 of the group.  sb_base: the super-block's 256 bytes (global memory or LDS).
__device__ __forceinline__ void q8k_store_class_major(signed char* sb_base, int lane, unsigned packed) {
  signed char* p = sb_base + 32 * (lane >> 3) + 16 * (lane & 1) + ((lane & 7) >> 1);
  p[0] = (signed char)(packed & 0xffu);
  p[4] = (signed char)((packed >> 8) & 0xffu);
  p[8] = (signed char)((packed >> 16) & 0xffu);
  p[12] = (signed char)(packed >> 24);
}

}  // namespace crabml_hip

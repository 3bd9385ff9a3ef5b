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
  asm volatile("" : "+v"(f));
  _Float16 x = (_Float16)f;
  unsigned short h;
  __builtin_memcpy(&h, &x, 2);
  return h;
}
// half: `a * b` / `a + b` on f16 = compute in f32, round once to f16
__device__ __forceinline__ unsigned short h_mul(unsigned short a, unsigned short b) { return f2h(h2f(a) * h2f(b)); }
__device__ __forceinline__ unsigned short h_add(unsigned short a, unsigned short b) { return f2h(h2f(a) + h2f(b)); }

__device__ __forceinline__ float wave_sum_f32(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ float wave_max_f32(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
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

}  // namespace crabml_hip

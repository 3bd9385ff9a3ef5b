// f16w_rows.hpp -- where an activation element sits in the pre-scaled f16 planes B' that the fast prompt pass's weight GEMM reads
// (gemm_f16w.hip: xh[row][k] halfs in the GEMM's k-slot order).  k_rows_to_f16 makes B' from finished planes; the row kernels of the
// pass that quantize a row (quantize.hip, prefill_rows.hpp) write it themselves with these, one launch fewer per GEMM.
#pragma once
#include "devutil.hpp"

namespace crabml_hip {

// order 0 (Q8_0 / Q8_1 rows; Q4_0, Q8_0, Q4_1 weights): slot of element j of a 32-element block -- the inverse of f16w_slot_elem
__device__ __forceinline__ int f16w_slot_of_elem(int j) {
  const int hi = j >> 4, s = (j >> 2) & 3, r = j & 3;
  return 8 * s + 4 * hi + ((r >> 1) | ((r & 1) << 1));
}
// orders 1 (Q4_K weights) / 2 (Q6_K weights), Q8_K rows: position (halfs from the row's start) of element E of super-block sb
__device__ __forceinline__ int f16w_pos_q8k(int order, int sb, int E) {
  int cc, g, s, hi, k;
  if (order == 1) {
    const int l = E & 7;
    g = E >> 6;
    hi = (E >> 5) & 1;
    k = (E >> 3) & 3;
    s = l & 3;
    cc = 2 * sb + (l >> 2);
  } else {
    const int r2 = E & 63;
    hi = (E >> 6) & 1;
    g = 2 * (r2 >> 5) + ((r2 >> 4) & 1);
    s = (r2 >> 2) & 3;
    k = r2 & 3;
    cc = 2 * sb + (E >> 7);
  }
  return (4 * cc + g) * 32 + 8 * s + 4 * hi + ((k >> 1) | ((k & 1) << 1));
}
// order 0: inside a block, slot 8 s + e (s = 0..3 = the dword of the weight block the slot pairs with) holds element
//   e = 0, 1: 4 s, 4 s + 2     e = 2, 3: 4 s + 1, 4 s + 3     e = 4, 5: 16 + 4 s, 16 + 4 s + 2     e = 6, 7: 16 + 4 s + 1, 16 + 4 s + 3
// -- the order in which unpack_q4_0_f16 takes the nibbles out of a dword (two masks per packed pair, no byte permute)
__device__ __forceinline__ int f16w_slot_elem(int slot) {
  const int s = slot >> 3, e = slot & 7;
  return (e >= 4 ? 16 : 0) + 4 * s + ((e >> 1) & 1) + 2 * (e & 1);
}
// one 8-slot group t = 4 kb + s of a row's B' from its FINISHED planes p (q | d at off_d): xrow[kb * 32 + 8 s ..]
template <int ORDER>
__device__ __forceinline__ void rows_to_f16_piece(const char* __restrict__ p, size_t off_d, int t, unsigned short* __restrict__ xrow) {
  const int kb = t >> 2, s = t & 3;
  float d;
  const signed char* q;
  int at[8];
  if constexpr (ORDER == 2) {
    const int cc = kb >> 2, g = kb & 3, sb = cc >> 1;
    d = ((const float*)(p + off_d))[sb];
    q = (const signed char*)p + sb * 256 + 128 * (cc & 1) + 32 * (g >> 1) + 16 * (g & 1) + 4 * s;
#pragma unroll
    for (int e = 0; e < 8; e++) at[e] = (e >= 4 ? 64 : 0) + ((e >> 1) & 1) + 2 * (e & 1);
  } else if constexpr (ORDER == 1) {
    const int cc = kb >> 2, g = kb & 3, sb = cc >> 1, l = 4 * (cc & 1) + s;
    d = ((const float*)(p + off_d))[sb];
    q = (const signed char*)p + sb * 256 + 64 * g;
#pragma unroll
    for (int e = 0; e < 8; e++) at[e] = (e >= 4 ? 32 : 0) + 8 * (((e >> 1) & 1) + 2 * (e & 1)) + l;
  } else {
    d = h2f(((const unsigned short*)(p + off_d))[kb]);
    q = (const signed char*)p + kb * 32;
#pragma unroll
    for (int e = 0; e < 8; e++) at[e] = f16w_slot_elem(8 * s + e);
  }
  unsigned short o[8];
#pragma unroll
  for (int e = 0; e < 8; e++) o[e] = f2h((float)q[at[e]] * d);  // (Q8_0: 7-bit x 11-bit, exact in f32, one rounding; Q8_K: f32 d, two)
  *(i32x4*)(xrow + kb * 32 + 8 * s) = i32x4{(int)(o[0] | ((unsigned)o[1] << 16)), (int)(o[2] | ((unsigned)o[3] << 16)),
                                            (int)(o[4] | ((unsigned)o[5] << 16)), (int)(o[6] | ((unsigned)o[7] << 16))};
}
// the f16 value of a quant: q * d in f32 (Q8_0 / Q8_1: d is the STORED f16 scale -- exact product, one rounding), rounded once
__device__ __forceinline__ unsigned short f16w_value(int q, float d) { return f2h((float)q * d); }

}  // namespace crabml_hip

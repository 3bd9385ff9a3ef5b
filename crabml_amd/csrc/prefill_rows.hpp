// prefill_rows.hpp -- row kernels of the FAST prompt pass that also leave the row's pre-scaled f16 plane B' for the weight GEMM that
// reads it next (gemm_f16w.hip; slot orders: f16w_rows.hpp): the same arithmetic as the kernels they wrap, one k_rows_to_f16 launch
// fewer per GEMM.  B' is bit for bit what k_rows_to_f16 makes from the finished planes.
#pragma once
#include "f16w_rows.hpp"
#include "fused_common.hpp"
#include "fused_ffn.hpp"

namespace crabml_hip {

// k_norm_quant_rows (residual add + RMSNorm + quantize, one workgroup per row; Q8_0 / Q8_1 planes) + the row's B' (order 0): the
// planes are read back by the workgroup that has just written them
// xh nullable.  parts / pstride / nparts: addv (the wo / ffn_down output that is about to be added to x) is piece 0 of a GEMM that was
// cut into k pieces (launch_gemm_f16w, defer_parts): the other pieces are added to it first, in piece order -- k_addn_f32's
// arithmetic, by the thread that reads the element next
template <int NIT, bool Q81>
__global__ __launch_bounds__(1024) void k_norm_quant_rows_h(float* __restrict__ x, float* __restrict__ addv, const float* __restrict__ w,
                                                           int cols, float eps, char* __restrict__ planes, size_t row_stride, size_t off_d,
                                                           size_t off_aux, int half, unsigned short* __restrict__ xh,
                                                           const float* __restrict__ parts, size_t pstride, int nparts) {
  extern __shared__ float lds[];
  __shared__ float s_rms;
  NormLds L{lds, lds + cols};
  const size_t r = blockIdx.x;
  char* p = planes + r * row_stride;
  if (nparts > 0) {
#pragma unroll
    for (int it = 0; it < NIT; it++) {
      const int i = it * 1024 + threadIdx.x;  // (norm_quant_block's own element -> thread mapping)
      if (i < cols) {
        float v = addv[r * cols + i];
        for (int s = 0; s < nparts; s++) v = v + parts[(size_t)s * pstride + r * cols + i];
        addv[r * cols + i] = v;
      }
    }
    __threadfence_block();
  }
  norm_quant_block<NIT, true, Q81>(x + r * cols, addv ? addv + r * cols : nullptr, w, cols, eps, L, &s_rms, (signed char*)p,
                                   (unsigned short*)(p + off_d), (void*)(p + off_aux), nullptr, half);
  if (xh == nullptr) return;
  __threadfence_block();
  __syncthreads();
  const int nb = cols / 32;
  for (int t = threadIdx.x; t < nb * 4; t += blockDim.x) rows_to_f16_piece<0>(p, off_d, t, xh + r * (size_t)cols);
}

// The same row (residual add + k pieces + RMSNorm + Q8_0 / Q8_1 quantize + B'), 256 threads instead of 1024: a thread owns E = cols / 256
// CONSECUTIVE elements -- 16 (half a 32-element chunk and quant block; cols = 4096) or 32 (a whole one; 8192) -- so the chunk's sum of
// squares, the block maximum and the block's integer sum are one register scan plus, for E = 16, one exchange with the neighbouring
// lane; nothing but the chunk sums goes through LDS (two barriers where norm_quant_block has four, and eight rows per CU in flight
// instead of two).  Arithmetic = norm_quant_block's, bit for bit: a chunk's squares from -0.0 in element order (half: rows 0..15 and
// 16..31 separately, then added -- the fast step's order; else one 32-element scan, rms_norm.rs:35-38); the chunk sums added as there
// (half: 64 per round through wave_sum_f32, rounds in order; else strictly in chunk order); (x / rms) * w; quant_lane32's quantizer
// (maximum and integer sum do not depend on the order).
template <int E, bool Q81>
__global__ __launch_bounds__(256) void k_norm_quant_rows_w(float* __restrict__ x, const float* __restrict__ addv, const float* __restrict__ w,
                                                          int cols, float eps, char* __restrict__ planes, size_t row_stride, size_t off_d,
                                                          size_t off_aux, int half, unsigned short* __restrict__ xh,
                                                          const float* __restrict__ parts, size_t pstride, int nparts) {
  static_assert(E == 16 || E == 32, "half a block or a whole one per thread");
  constexpr int V = E / 4;
  __shared__ float cs_lds[256];
  __shared__ float s_rms;
  const size_t r = blockIdx.x;
  const int tid = threadIdx.x;
  const size_t e0 = r * (size_t)cols + (size_t)E * tid;
  f32x4 xv[V], wv[V];
#pragma unroll
  for (int j = 0; j < V; j++) {
    xv[j] = ((const f32x4*)(x + e0))[j];
    wv[j] = ((const f32x4*)(w + (size_t)E * tid))[j];
  }
  if (addv != nullptr) {
#pragma unroll
    for (int j = 0; j < V; j++) {
      f32x4 a = ((const f32x4*)(addv + e0))[j];
      for (int s = 0; s < nparts; s++) a = a + ((const f32x4*)(parts + (size_t)s * pstride + e0))[j];  // (k_addn_f32's order)
      xv[j] = a + xv[j];  // x = matmul_out + x (llama2.rs:266 / :636)
      ((f32x4*)(x + e0))[j] = xv[j];
    }
  }
  // ---- the chunk's sum of squares
  float cs;
  if constexpr (E == 32) {
    float s0 = -0.0f, s1 = -0.0f;
#pragma unroll
    for (int j = 0; j < V; j++)
#pragma unroll
      for (int i = 0; i < 4; i++) {
        float& a = (half && j >= 4) ? s1 : s0;
        a += xv[j][i] * xv[j][i];
      }
    cs = half ? s0 + s1 : s0;
    cs_lds[tid] = cs;
  } else {
    float hs = -0.0f;
#pragma unroll
    for (int j = 0; j < V; j++)
#pragma unroll
      for (int i = 0; i < 4; i++) hs += xv[j][i] * xv[j][i];
    const float other = dpp_f<0xB1>(hs);  // the neighbouring lane's (lane ^ 1)
    if (half) {
      cs = (tid & 1) ? other + hs : hs + other;  // rows 0..15 + rows 16..31
    } else {
      float sc = other;  // the odd lane continues the even lane's scan
#pragma unroll
      for (int j = 0; j < V; j++)
#pragma unroll
        for (int i = 0; i < 4; i++) sc += xv[j][i] * xv[j][i];
      cs = sc;  // (meaningful in the odd lane)
    }
    if ((tid & 1) == 1) cs_lds[tid >> 1] = cs;
  }
  __syncthreads();
  const int nchunks = cols / 32;
  if (tid < 64) {
    float sum = 0.0f;
    for (int base = 0; base < nchunks; base += 64) {
      const float v = base + tid < nchunks ? cs_lds[base + tid] : 0.0f;
      if (half) {
        sum += wave_sum_f32(v);
      } else {
#pragma unroll
        for (int i = 0; i < 64; i++) sum += rl_f(v, i);
      }
    }
    if (tid == 0) s_rms = sqrtf(sum / (float)cols + eps);
  }
  __syncthreads();
  const float rms = s_rms;
  // ---- normalize, quantize (a 32-element block = this thread, or this thread and lane ^ 1)
  float amax = 0.0f;
#pragma unroll
  for (int j = 0; j < V; j++)
#pragma unroll
    for (int i = 0; i < 4; i++) {
      xv[j][i] = (xv[j][i] / rms) * wv[j][i];
      amax = fmaxf(amax, fabsf(xv[j][i]));
    }
  if constexpr (E == 16) amax = fmaxf(amax, dpp_f<0xB1>(amax));
  const float dd = amax / 127.0f;
  const unsigned short dh = f2h(dd);
  int q[E], sum = 0;
#pragma unroll
  for (int e = 0; e < E; e++) {
    const float v = xv[e >> 2][e & 3];
    if constexpr (Q81)
      q[e] = (int)fminf(fmaxf(v / dd, -128.0f), 127.0f);
    else
      q[e] = (int)(signed char)(unsigned char)((unsigned)rs_f32_as_i32(v / dd) & 0xffu);
    sum += q[e];
  }
  if constexpr (E == 16) sum += dpp_i<0xB1>(sum);
  char* p = planes + r * row_stride;
#pragma unroll
  for (int j = 0; j < E / 16; j++) {
    i32x4 pk;
#pragma unroll
    for (int i = 0; i < 4; i++)
      pk[i] = (int)(((unsigned)q[16 * j + 4 * i] & 0xffu) | (((unsigned)q[16 * j + 4 * i + 1] & 0xffu) << 8) |
                    (((unsigned)q[16 * j + 4 * i + 2] & 0xffu) << 16) | (((unsigned)q[16 * j + 4 * i + 3] & 0xffu) << 24));
    ((i32x4*)(p + (size_t)E * tid))[j] = pk;
  }
  const int blk = E == 32 ? tid : tid >> 1;
  if (E == 32 || (tid & 1) == 0) {
    ((unsigned short*)(p + off_d))[blk] = dh;
    if constexpr (Q81)
      ((unsigned short*)(p + off_aux))[blk] = f2h((float)sum * dd);
    else
      ((int*)(p + off_aux))[blk] = sum;
  }
  if (xh != nullptr) {
    const float ds = h2f(dh);
    unsigned short* xr = xh + r * (size_t)cols + (size_t)blk * 32;
#pragma unroll
    for (int hj = 0; hj < E / 16; hj++) {
      const int hi = E == 32 ? hj : (tid & 1);  // which 16 elements of the block
#pragma unroll
      for (int s4 = 0; s4 < 4; s4++) {  // elements 16 hi + 4 s4 + {0, 2, 1, 3} -> slots 8 s4 + 4 hi + {0, 1, 2, 3} (f16w_slot_of_elem)
        const int b = 16 * hj + 4 * s4;
        const unsigned lo = (unsigned)f16w_value(q[b], ds) | ((unsigned)f16w_value(q[b + 2], ds) << 16);
        const unsigned hi2 = (unsigned)f16w_value(q[b + 1], ds) | ((unsigned)f16w_value(q[b + 3], ds) << 16);
        *(unsigned long long*)(xr + 8 * s4 + 4 * hi) = (unsigned long long)lo | ((unsigned long long)hi2 << 32);
      }
    }
  }
}

// Q8_K rows (K-quant layers): residual add (+ the k pieces of the GEMM that made it) + RMSNorm (k_norm_f32_rows' arithmetic: xn goes to
// memory as there) + the Q8_K quantizer (k_quantize_q8_k's: a wave per super-block, on the xn the workgroup has just written) + the
// row's B' in the k-slot order of the weight format that reads it -- one launch where the pass had k_addn_f32, k_res_epi, k_norm_f32_rows
// and k_quantize_q8_k
template <int NIT>
__global__ __launch_bounds__(1024) void k_norm_quant_rows_k(float* __restrict__ x, float* __restrict__ addv, const float* __restrict__ w, int cols,
                                                           float eps, float* __restrict__ xn, char* __restrict__ planes, size_t row_stride,
                                                           size_t off_d, size_t off_aux, size_t off_p, int half, unsigned short* __restrict__ xh,
                                                           int xh_order, const float* __restrict__ parts, size_t pstride, int nparts) {
  extern __shared__ float lds[];
  __shared__ float s_rms;
  NormLds L{lds, lds + cols};
  const size_t r = blockIdx.x;
  if (nparts > 0) {
#pragma unroll
    for (int it = 0; it < NIT; it++) {
      const int i = it * 1024 + threadIdx.x;  // (norm_quant_block's own element -> thread mapping)
      if (i < cols) {
        float v = addv[r * cols + i];
        for (int s = 0; s < nparts; s++) v = v + parts[(size_t)s * pstride + r * cols + i];
        addv[r * cols + i] = v;
      }
    }
    __threadfence_block();
  }
  float* xr = xn + r * cols;
  norm_quant_block<NIT, false>(x + r * cols, addv ? addv + r * cols : nullptr, w, cols, eps, L, &s_rms, nullptr, nullptr, nullptr, xr, half);
  __threadfence_block();
  __syncthreads();
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nsb = cols / 256;
  char* p = planes + r * row_stride;
  for (int sb = wave; sb < nsb; sb += 16) {
    const f32x4 v = ((const f32x4*)xr)[sb * 64 + lane];
    const Q8KLane o = q8k_wave_quant(v, lane);
    *(unsigned*)(p + sb * 256 + lane * 4) = o.packed;
    q8k_store_class_major((signed char*)(p + off_p) + sb * 256, lane, o.packed);
    if ((lane & 3) == 0) ((short*)(p + off_aux))[sb * 16 + (lane >> 2)] = (short)o.quad_sum;
    if (lane == 0) ((float*)(p + off_d))[sb] = o.d;
    if (xh) {
      unsigned short* xo = xh + r * (size_t)cols;
#pragma unroll
      for (int i = 0; i < 4; i++)
        xo[f16w_pos_q8k(xh_order, sb, 4 * lane + i)] = f16w_value((int)(signed char)((o.packed >> (8 * i)) & 0xffu), o.d);
    }
  }
}

// k_gateup_epi_quant (h = silu(g) * u quantized straight into the rows' Q8_0 / Q8_1 planes) + the rows' B' (order 0)
template <bool Q81>
__global__ __launch_bounds__(256) void k_gateup_epi_quant_h(const float* __restrict__ g, const float* __restrict__ u,
                                                            const unsigned short* __restrict__ exp_tab, int hidden, char* __restrict__ planes,
                                                            size_t row_stride, size_t off_d, size_t off_aux, unsigned short* __restrict__ xh) {
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
  xh[r * (size_t)hidden + (i & ~31) + f16w_slot_of_elem(i & 31)] = f16w_value((int)o.q, h2f(o.d));
}

}  // namespace crabml_hip

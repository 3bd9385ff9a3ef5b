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

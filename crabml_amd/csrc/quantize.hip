// quantize.hip -- on-the-fly activation quantization for matmul_vec's rhs.
//
// Replaces CpuTensorBuf::quantize (crabml-core/src/cpu/buf/api.rs:195-228) as called from
// gemv_dense_2d_2d (crabml-core/src/cpu/primitives/matmul_vec.rs:37-40).  The arithmetic follows the
// reference, NOT ggml:
//   Q8_0  d = max|x| / 127 ; q = trunc(x / d)  (true division; simd cast: NaN -> 0)   buf_q8_0.rs:87-134
//   Q8_1  d = max|x| / 127 ; q = trunc(clamp(x / d, -128, 127)) (NaN -> -128) ; s = f16(d * sum q)
//                                                                                    buf_q8_1.rs:90-129
//   Q8_K  mx = first element of max |x| ; scale = -128 / mx ; q = min(round_half_away(scale * x), 127) ;
//         d = 1 / scale ; bsums over 16                                              buf_q8_k.rs:84-131
// Output goes to "planes" (see common.hpp).  All outputs are bit-identical to the reference's blocks
// (tests/test_hip_quantize.py compares the bytes).
#include "devutil.hpp"
#include "f16w_rows.hpp"
#include "kernels.hpp"

namespace crabml_hip {

// one 32-lane group per block; 256 threads = 8 blocks per workgroup
__global__ __launch_bounds__(256) void k_quantize_q8_0(const float* __restrict__ x, signed char* __restrict__ q,
                                                       unsigned short* __restrict__ d, int* __restrict__ isum,
                                                       size_t nblocks, size_t row_elems, size_t row_bytes, unsigned short* __restrict__ xh) {
  // blockIdx.y: row of a batch (x rows of row_elems floats; one set of planes every row_bytes bytes)
  // xh (nullable): the row's pre-scaled f16 plane for the fast prompt pass's GEMM (f16w_rows.hpp), written next to the planes
  x += blockIdx.y * row_elems;
  q += blockIdx.y * row_bytes;
  d = (unsigned short*)((char*)d + blockIdx.y * row_bytes);
  isum = (int*)((char*)isum + blockIdx.y * row_bytes);
  size_t gid = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  size_t blk = gid >> 5;
  int j = (int)(gid & 31);
  bool live = blk < nblocks;
  float v = live ? x[blk * 32 + j] : 0.f;
  float amax = half_max_f32(fabsf(v));  // (256-thread blocks of whole waves: all lanes converged)
  float dd = amax / 127.0f;
  float t = v / dd;
  int qi = rs_f32_as_i32(t);
  signed char q8 = (signed char)(unsigned char)((unsigned)qi & 0xffu);  // `as i8` from i32 wraps
  int s = half_sum_i32((int)q8);
  if (live) {
    q[blk * 32 + j] = q8;
    if (j == 0) {
      d[blk] = f2h(dd);
      isum[blk] = s;
    }
    if (xh) xh[blockIdx.y * row_elems + blk * 32 + f16w_slot_of_elem(j)] = f16w_value((int)q8, h2f(f2h(dd)));
  }
}

__global__ __launch_bounds__(256) void k_quantize_q8_1(const float* __restrict__ x, signed char* __restrict__ q,
                                                       unsigned short* __restrict__ d, unsigned short* __restrict__ sp,
                                                       size_t nblocks, size_t row_elems, size_t row_bytes, unsigned short* __restrict__ xh) {
  x += blockIdx.y * row_elems;
  q += blockIdx.y * row_bytes;
  d = (unsigned short*)((char*)d + blockIdx.y * row_bytes);
  sp = (unsigned short*)((char*)sp + blockIdx.y * row_bytes);
  size_t gid = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  size_t blk = gid >> 5;
  int j = (int)(gid & 31);
  bool live = blk < nblocks;
  float v = live ? x[blk * 32 + j] : 0.f;
  float amax = half_max_f32(fabsf(v));
  float dd = amax / 127.0f;
  float sv = v / dd;
  // Rust f32::max / f32::min return the non-NaN operand: NaN.max(-128) = -128
  float c = fminf(fmaxf(sv, -128.0f), 127.0f);
  int qi = (int)c;  // |c| <= 128: exact truncation
  int s = half_sum_i32(qi);
  if (live) {
    q[blk * 32 + j] = (signed char)qi;
    if (j == 0) {
      d[blk] = f2h(dd);
      // s accumulates small integers in f32 in the reference (exact), then `s *= d`
      sp[blk] = f2h((float)s * dd);
    }
    if (xh) xh[blockIdx.y * row_elems + blk * 32 + f16w_slot_of_elem(j)] = f16w_value(qi, h2f(f2h(dd)));
  }
}

// one wave per 256-element super-block, 4 consecutive elements per lane
__global__ __launch_bounds__(256) void k_quantize_q8_k(const float* __restrict__ x, signed char* __restrict__ q,
                                                       float* __restrict__ d, short* __restrict__ bsums, signed char* __restrict__ qp,
                                                       size_t nblocks, size_t row_elems, size_t row_bytes, unsigned short* __restrict__ xh, int xh_order) {
  x += blockIdx.y * row_elems;
  q += blockIdx.y * row_bytes;
  qp += blockIdx.y * row_bytes;
  d = (float*)((char*)d + blockIdx.y * row_bytes);
  bsums = (short*)((char*)bsums + blockIdx.y * row_bytes);
  const int lane = threadIdx.x & 63;
  size_t blk = (size_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
  if (blk >= nblocks) return;  // whole wave exits together
  const f32x4 v = *(const f32x4*)(x + blk * 256 + lane * 4);
  const Q8KLane o = q8k_wave_quant(v, lane);  // devutil.hpp
  *(unsigned*)(q + blk * 256 + lane * 4) = o.packed;
  q8k_store_class_major(qp + blk * 256, lane, o.packed);  // the plane the Q4_K kernels read
  if ((lane & 3) == 0) bsums[blk * 16 + (lane >> 2)] = (short)o.quad_sum;
  if (lane == 0) d[blk] = o.d;
  if (xh) {  // (xh_order 1 / 2: the k-slot order of Q4_K / Q6_K weights)
    unsigned short* xr = xh + blockIdx.y * row_elems;
#pragma unroll
    for (int i = 0; i < 4; i++)
      xr[f16w_pos_q8k(xh_order, (int)blk, 4 * lane + i)] = f16w_value((int)(signed char)((o.packed >> (8 * i)) & 0xffu), o.d);
  }
}

__global__ __launch_bounds__(256) void k_quantize_f16(const float* __restrict__ x, unsigned short* __restrict__ h,
                                                      size_t n, size_t row_bytes) {
  x += blockIdx.y * n;
  h = (unsigned short*)((char*)h + blockIdx.y * row_bytes);
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) h[i] = f2h(x[i]);
}

void launch_quantize_act(hipStream_t st, uint32_t qtype, const float* x, size_t n, void* planes) {
  launch_quantize_act_rows(st, qtype, x, 1, n, planes);
}
// xh (nullable; Q8_0 / Q8_1 / Q8_K): (rows, n) halfs -- the rows' pre-scaled f16 planes in k-slot order `xh_order` (gemm_f16w_order of the
// weight format that will read them), bit for bit what launch_rows_to_f16 makes from the finished planes
// rows vectors of n elements each -> rows sets of planes, act_layout(qtype, n).total bytes apart (the batched rhs of
// launch_gemv)
void launch_quantize_act_rows(hipStream_t st, uint32_t qtype, const float* x, size_t rows, size_t n, void* planes, void* xh, int xh_order) {
  if (n == 0 || rows == 0) return;
  ActLayout al = act_layout(qtype, n);
  const unsigned ry = (unsigned)rows;
  char* p = (char*)planes;
  switch (qtype) {
    case CRABML_HIP_Q8_0: {
      size_t nb = n / 32;
      unsigned grid = (unsigned)((nb * 32 + 255) / 256);
      k_quantize_q8_0<<<dim3(grid, ry), 256, 0, st>>>(x, (signed char*)p, (unsigned short*)(p + al.off_d), (int*)(p + al.off_aux),
                                           nb, n, al.total, (unsigned short*)xh);
      break;
    }
    case CRABML_HIP_Q8_1: {
      size_t nb = n / 32;
      unsigned grid = (unsigned)((nb * 32 + 255) / 256);
      k_quantize_q8_1<<<dim3(grid, ry), 256, 0, st>>>(x, (signed char*)p, (unsigned short*)(p + al.off_d),
                                           (unsigned short*)(p + al.off_aux), nb, n, al.total, (unsigned short*)xh);
      break;
    }
    case CRABML_HIP_Q8_K: {
      size_t nb = n / 256;
      unsigned grid = (unsigned)((nb + 3) / 4);
      k_quantize_q8_k<<<dim3(grid, ry), 256, 0, st>>>(x, (signed char*)p, (float*)(p + al.off_d), (short*)(p + al.off_aux),
                                                      (signed char*)(p + al.off_p), nb, n, al.total, (unsigned short*)xh, xh_order);
      break;
    }
    case CRABML_HIP_F16: {
      unsigned grid = (unsigned)((n + 255) / 256);
      k_quantize_f16<<<dim3(grid, ry), 256, 0, st>>>(x, (unsigned short*)p, n, al.total);
      break;
    }
    default: break;
  }
}

}  // namespace crabml_hip

// attention.hip -- Tensor::batch_matmul: the KV-cache attention dot (Q.K^T) and combine (P.V).
//
// Replaces crabml-core/src/cpu/primitives/batch_matmul.rs:15-131 with the reference's exact arithmetic:
//   B = F32 cache  : c += a * b, k ascending, f32 (batch_matmul_naive_f32, :47-71); B batch = bi % bb.
//   B = F16 cache  : A is rounded to f16 first (:39); B batch = bi / (ba/bb)  (GQA broadcast, :89-91)
//      stride_k==1 : QK^T, f32 accumulation of f16*f16 products, k ascending (buf_f16.rs:83-97)
//      stride_n==1 : PV, accumulated IN f16 -- every step rounds the product and the sum to f16
//                    (vec_fma_f16_f16, buf_f16.rs:152-163).  Reproduced exactly, not "improved".
// One thread per output element with the k loop sequential keeps the reference's association, so the
// result is bit-identical; n (the contiguous axis of P.V) maps to lanes so V rows are read coalesced.
#include "devutil.hpp"
#include "kernels.hpp"

namespace crabml_hip {

__global__ __launch_bounds__(256) void k_bmm_f32(const float* __restrict__ a, size_t ba, size_t m, size_t k,
                                                 const float* __restrict__ b, size_t bb, size_t n, size_t sb0,
                                                 size_t sb1, size_t sb2, float* __restrict__ c) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= ba * m * n) return;
  size_t ni = i % n, mi = (i / n) % m, bi = i / (m * n);
  const float* ar = a + bi * (m * k) + mi * k;
  const float* br = b + (bi % bb) * sb0 + ni * sb2;
  float acc = 0.0f;
  for (size_t ki = 0; ki < k; ki++) acc += ar[ki] * br[ki * sb1];
  c[i] = acc;
}

// stride_k == 1
__global__ __launch_bounds__(256) void k_bmm_f16_dot(const float* __restrict__ a, size_t ba, size_t m, size_t k,
                                                     const unsigned short* __restrict__ b, size_t bb, size_t n,
                                                     size_t sb0, size_t sb2, float* __restrict__ c) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= ba * m * n) return;
  size_t ni = i % n, mi = (i / n) % m, bi = i / (m * n);
  size_t bcast = ba / bb;
  const float* ar = a + bi * (m * k) + mi * k;
  const unsigned short* br = b + (bi / bcast) * sb0 + ni * sb2;
  float acc = 0.0f;
  for (size_t ki = 0; ki < k; ki++) acc += h2f(f2h(ar[ki])) * h2f(br[ki]);
  c[i] = acc;
}

// stride_n == 1
__global__ __launch_bounds__(256) void k_bmm_f16_fma(const float* __restrict__ a, size_t ba, size_t m, size_t k,
                                                     const unsigned short* __restrict__ b, size_t bb, size_t n,
                                                     size_t sb0, size_t sb1, float* __restrict__ c) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= ba * m * n) return;
  size_t ni = i % n, mi = (i / n) % m, bi = i / (m * n);
  size_t bcast = ba / bb;
  const float* ar = a + bi * (m * k) + mi * k;
  const unsigned short* bc = b + (bi / bcast) * sb0 + ni;
  _Float16 acc = (_Float16)0.0f;  // f16::ZERO
  for (size_t ki = 0; ki < k; ki++) {
    const _Float16 prod = hbits(bc[ki * sb1]) * hbits(f2h(ar[ki]));  // native f16 ops: see devutil.hpp
    acc = acc + prod;
  }
  c[i] = (float)acc;
}

void launch_batch_matmul(hipStream_t st, const float* a, size_t ba, size_t m, size_t k, const void* b, int b_f16,
                         size_t bb, size_t n, size_t sb0, size_t sb1, size_t sb2, float* c) {
  size_t total = ba * m * n;
  if (total == 0) return;
  unsigned grid = (unsigned)((total + 255) / 256);
  if (!b_f16)
    k_bmm_f32<<<grid, 256, 0, st>>>(a, ba, m, k, (const float*)b, bb, n, sb0, sb1, sb2, c);
  else if (sb1 == 1)
    k_bmm_f16_dot<<<grid, 256, 0, st>>>(a, ba, m, k, (const unsigned short*)b, bb, n, sb0, sb2, c);
  else
    k_bmm_f16_fma<<<grid, 256, 0, st>>>(a, ba, m, k, (const unsigned short*)b, bb, n, sb0, sb1, c);
}

}  // namespace crabml_hip

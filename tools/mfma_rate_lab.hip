// mfma_rate_lab.hip -- issue rate of the matrix-core instructions the prompt GEMMs can use (one wave per SIMD and four waves per
// SIMD, independent accumulators, no memory traffic): cycles per instruction per SIMD.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -o build/mfma_rate_lab tools/mfma_rate_lab.hip
#include <hip/hip_runtime.h>

#include <cstdio>
typedef int i32x4 __attribute__((ext_vector_type(4)));
typedef int i32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 h16x8 __attribute__((ext_vector_type(8)));
#define CK(x)                                                                                \
  do {                                                                                       \
    hipError_t e_ = (x);                                                                     \
    if (e_ != hipSuccess) {                                                                  \
      fprintf(stderr, "HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); \
      exit(1);                                                                               \
    }                                                                                        \
  } while (0)

template <int MODE>
__global__ void k_rate(const int* __restrict__ in, int* __restrict__ out, int n) {
  const long a8 = ((const long*)in)[threadIdx.x & 63];
  const i32x4 a16 = ((const i32x4*)in)[threadIdx.x & 63];
  const h16x8 ah = __builtin_bit_cast(h16x8, a16);
  i32x4 c0 = {0, 0, 0, 0}, c1 = c0, c2 = c0, c3 = c0;
  f32x4 f0 = {0, 0, 0, 0}, f1 = f0, f2 = f0, f3 = f0;
  for (int it = 0; it < n; it += 4) {
    if (MODE == 0) {  // v_mfma_i32_16x16x32_i8
      c0 = __builtin_amdgcn_mfma_i32_16x16x32_i8(a8, a8, c0, 0, 0, 0);
      c1 = __builtin_amdgcn_mfma_i32_16x16x32_i8(a8, a8, c1, 0, 0, 0);
      c2 = __builtin_amdgcn_mfma_i32_16x16x32_i8(a8, a8, c2, 0, 0, 0);
      c3 = __builtin_amdgcn_mfma_i32_16x16x32_i8(a8, a8, c3, 0, 0, 0);
    } else if (MODE == 1) {  // v_mfma_i32_16x16x64_i8
      c0 = __builtin_amdgcn_mfma_i32_16x16x64_i8(a16, a16, c0, 0, 0, 0);
      c1 = __builtin_amdgcn_mfma_i32_16x16x64_i8(a16, a16, c1, 0, 0, 0);
      c2 = __builtin_amdgcn_mfma_i32_16x16x64_i8(a16, a16, c2, 0, 0, 0);
      c3 = __builtin_amdgcn_mfma_i32_16x16x64_i8(a16, a16, c3, 0, 0, 0);
    } else {  // v_mfma_f32_16x16x32_f16
      f0 = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, ah, f0, 0, 0, 0);
      f1 = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, ah, f1, 0, 0, 0);
      f2 = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, ah, f2, 0, 0, 0);
      f3 = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, ah, f3, 0, 0, 0);
    }
  }
  c0 = c0 + c1 + c2 + c3;
  f0 = f0 + f1 + f2 + f3;
  out[blockIdx.x * blockDim.x + threadIdx.x] = c0[0] + c0[1] + c0[2] + c0[3] + (int)(f0[0] + f0[1] + f0[2] + f0[3]);
}

// one MFMA (fresh accumulator) + NV f32 operations on its four results per iteration: how do the two pipes share a SIMD?
template <int NV>
__global__ void k_mix(const int* __restrict__ in, int* __restrict__ out, int n) {
  const long a8 = ((const long*)in)[threadIdx.x & 63];
  const float s0 = __builtin_bit_cast(float, in[(threadIdx.x + 1) & 63]) + 1.0f, s1 = s0 + 0.5f;
  const i32x4 CM = {0x4B400000, 0x4B400000, 0x4B400000, 0x4B400000};
  float F[4] = {0.f, 0.f, 0.f, 0.f};
  for (int it = 0; it < n; it++) {
    const i32x4 D = __builtin_amdgcn_mfma_i32_16x16x32_i8(a8 + it, a8, CM, 0, 0, 0);
    const f32x4 Df = __builtin_bit_cast(f32x4, D);
#pragma unroll
    for (int r = 0; r < 4; r++) {
      float t = Df[r];
      if (NV >= 4) t = __builtin_fmaf(t, s0, -12582912.0f * s0);
      if (NV >= 8) t = t * s1;
      if (NV >= 12) F[r] += t;
      else F[r] = __builtin_bit_cast(float, __builtin_bit_cast(int, F[r]) ^ __builtin_bit_cast(int, t));
      if (NV >= 16) F[r] = F[r] * s1;
    }
  }
  out[blockIdx.x * blockDim.x + threadIdx.x] = (int)(F[0] + F[1] + F[2] + F[3]);
}

int main() {
  int *in, *out;
  CK(hipMalloc(&in, 64 * 16));
  CK(hipMemset(in, 0, 64 * 16));
  CK(hipMalloc(&out, 256 * 1024 * 4));
  const int n = 1 << 16;
  const char* names[] = {"v_mfma_i32_16x16x32_i8", "v_mfma_i32_16x16x64_i8", "v_mfma_f32_16x16x32_f16"};
  auto run = [&](int mode, auto kern, int threads, int blocks) {
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    kern<<<blocks, threads>>>(in, out, n);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0));
    kern<<<blocks, threads>>>(in, out, n);
    CK(hipEventRecord(e1));
    CK(hipDeviceSynchronize());
    float ms;
    CK(hipEventElapsedTime(&ms, e0, e1));
    const int waves_per_simd = threads / 256;
    const double cyc = ms * 1e-3 * 2.4e9 / ((double)n * waves_per_simd);
    printf("%-26s %d wave(s) per SIMD, %4d workgroups: %6.2f cycles per instruction per SIMD (at 2.4 GHz)\n", names[mode], waves_per_simd, blocks,
           cyc);
  };
  for (int blocks : {1, 256}) {
    run(0, k_rate<0>, 256, blocks);
    run(0, k_rate<0>, 1024, blocks);
    run(1, k_rate<1>, 256, blocks);
    run(1, k_rate<1>, 1024, blocks);
    run(2, k_rate<2>, 256, blocks);
    run(2, k_rate<2>, 1024, blocks);
  }
  printf("\none v_mfma_i32_16x16x32_i8 + NV f32 operations on its results per iteration (cycles per iteration per SIMD):\n");
  auto runmix = [&](int nv, auto kern, int threads) {
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    kern<<<256, threads>>>(in, out, n);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0));
    kern<<<256, threads>>>(in, out, n);
    CK(hipEventRecord(e1));
    CK(hipDeviceSynchronize());
    float ms;
    CK(hipEventElapsedTime(&ms, e0, e1));
    const int w = threads / 256;
    printf("  NV = %2d (+4 xor when NV < 12), %d wave(s) per SIMD: %7.2f cycles per wave-iteration, %7.2f per SIMD\n", nv, w, ms * 1e-3 * 2.4e9 / n,
           ms * 1e-3 * 2.4e9 / ((double)n * w));
  };
  for (int threads : {256, 512, 1024}) {
    runmix(0, k_mix<0>, threads);
    runmix(4, k_mix<4>, threads);
    runmix(8, k_mix<8>, threads);
    runmix(12, k_mix<12>, threads);
    runmix(16, k_mix<16>, threads);
  }
  return 0;
}

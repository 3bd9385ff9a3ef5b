// valu_chain_lab.hip -- dependent-issue latency of the f16 adds the attention PV chain is made of (one wave on an idle CU).
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -o build/valu_chain_lab tools/valu_chain_lab.hip
#include <hip/hip_runtime.h>

#include <cstdio>
typedef _Float16 h16x2 __attribute__((ext_vector_type(2)));
#define CK(x)                                                                                \
  do {                                                                                       \
    hipError_t e_ = (x);                                                                     \
    if (e_ != hipSuccess) {                                                                  \
      fprintf(stderr, "HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); \
      exit(1);                                                                               \
    }                                                                                        \
  } while (0)

// MODE 0: one v_pk_add_f16 chain; 1: two independent v_pk_add_f16 chains; 2: one v_add_f16 chain; 3: two v_add_f16 chains;
// 4: one v_add_f32 chain; 5: four v_pk_add_f16 chains; 6: v_pk_mul_f16 (independent) + dependent v_pk_add_f16
template <int MODE>
__global__ void k_chain(const unsigned* __restrict__ in, unsigned* __restrict__ out, long long* __restrict__ cyc, int n) {
  unsigned x[16];
#pragma unroll
  for (int i = 0; i < 16; i++) x[i] = in[threadIdx.x * 16 + i];
  h16x2 c0 = {0, 0}, c1 = {0, 0}, c2 = {0, 0}, c3 = {0, 0};
  _Float16 s0 = 0, s1 = 0;
  float f0 = 0;
  const long long t0 = __builtin_readcyclecounter();
  for (int it = 0; it < n; it += 16) {
#pragma unroll
    for (int i = 0; i < 16; i++) {
      const h16x2 v = __builtin_bit_cast(h16x2, x[i]);
      if (MODE == 0) c0 = c0 + v;
      if (MODE == 1) { c0 = c0 + v; c1 = c1 + v; }
      if (MODE == 2) s0 = s0 + v[0];
      if (MODE == 3) { s0 = s0 + v[0]; s1 = s1 + v[1]; }
      if (MODE == 4) f0 = f0 + __builtin_bit_cast(float, x[i]);
      if (MODE == 5) { c0 = c0 + v; c1 = c1 + v; c2 = c2 + v; c3 = c3 + v; }
      if (MODE == 6) c0 = c0 + v * __builtin_bit_cast(h16x2, x[(i + 1) & 15]);
      if (MODE == 7) { s0 = s0 + v[0]; s0 = s0 + v[1]; }  // low half, then high half (SDWA source select)
    }
    asm volatile("" ::: "memory");
  }
  const long long t1 = __builtin_readcyclecounter();
  c0 = c0 + c1 + c2 + c3;
  out[threadIdx.x] = __builtin_bit_cast(unsigned, c0) ^ (unsigned)__builtin_bit_cast(unsigned short, s0) ^
                     ((unsigned)__builtin_bit_cast(unsigned short, s1) << 16) ^ __builtin_bit_cast(unsigned, f0);
  if (threadIdx.x == 0) *cyc = t1 - t0;
}

int main() {
  unsigned *in, *out;
  long long* cyc;
  CK(hipMalloc(&in, 64 * 16 * 4));
  CK(hipMemset(in, 0, 64 * 16 * 4));
  CK(hipMalloc(&out, 64 * 4));
  CK(hipMalloc(&cyc, 8));
  const int n = 1 << 16;
  const char* names[] = {"one v_pk_add_f16 chain", "two v_pk_add_f16 chains", "one v_add_f16 chain", "two v_add_f16 chains",
                         "one v_add_f32 chain", "four v_pk_add_f16 chains", "v_pk_mul_f16 + dependent v_pk_add_f16",
                         "one v_add_f16 chain, low / high halves (2 steps)"};
  auto run = [&](int mode, auto kern) {
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    kern<<<1, 64>>>(in, out, cyc, n);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0));
    kern<<<1, 64>>>(in, out, cyc, n);
    CK(hipEventRecord(e1));
    CK(hipDeviceSynchronize());
    float ms;
    CK(hipEventElapsedTime(&ms, e0, e1));
    long long c;
    CK(hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost));
    printf("%-44s %6.2f ns per step (%5.2f cycles at 2.4 GHz), s_memtime ticks per step %.3f\n", names[mode], ms * 1e6 / n, ms * 1e6 / n * 2.4,
           (double)c / n);
  };
  run(0, k_chain<0>);
  run(1, k_chain<1>);
  run(2, k_chain<2>);
  run(3, k_chain<3>);
  run(4, k_chain<4>);
  run(5, k_chain<5>);
  run(6, k_chain<6>);
  run(7, k_chain<7>);
  return 0;
}

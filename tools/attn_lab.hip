// attn_lab.hip -- where the decode attention kernel (k_attn, one workgroup per head) spends its time at short context.
// Not part of the product: it includes the product's kernel header and launches k_attn<true, STAMP> on a synthetic f16 KV
// cache of the Llama-3-8B geometry (32 heads, 8 kv heads, head_dim 128); workgroup 0 stamps s_memtime at its phase
// boundaries.  Also times the launch with / without the 224 spare workgroups that prefetch the wo weights.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -I include -o build/attn_lab tools/attn_lab.hip
#include "../crabml_amd/csrc/fused_attention.hpp"

#include <algorithm>
#include <cstdio>
#include <vector>

using namespace crabml_hip;
#define CK(x)                                                                                \
  do {                                                                                       \
    hipError_t e_ = (x);                                                                     \
    if (e_ != hipSuccess) {                                                                  \
      fprintf(stderr, "HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); \
      exit(1);                                                                               \
    }                                                                                        \
  } while (0)

// stubs for the symbols common.hpp declares (the lab links nothing from the library)
namespace crabml_hip {
int set_error(crabml_hip_device*, int status, const char*, ...) { return status; }
int hip_fail(crabml_hip_device*, hipError_t, const char*, const char*, int) { return 1; }
}  // namespace crabml_hip

int main() {
  const int n_heads = 32, n_kv = 8, hd = 128, seq_cap = 8192;
  const int L = getenv("ATTN_LAB_L") ? atoi(getenv("ATTN_LAB_L")) : 24;  // L cache copies: every launch touches cold-ish rows (L = 1: warm)
  printf("# %d KV-cache copies\n", L);
  hipStream_t st;
  CK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
  const size_t kvb = (size_t)n_kv * seq_cap * hd * 2;
  char *kc, *vc;
  CK(hipMalloc(&kc, kvb * L));
  CK(hipMalloc(&vc, kvb * L));
  std::vector<unsigned short> h(kvb / 2);
  for (size_t i = 0; i < h.size(); i++) h[i] = (unsigned short)(0x3000 + (i * 2654435761u >> 20) % 0x0800);  // small positive f16
  for (int l = 0; l < L; l++) {
    CK(hipMemcpy(kc + l * kvb, h.data(), kvb, hipMemcpyHostToDevice));
    CK(hipMemcpy(vc + l * kvb, h.data(), kvb, hipMemcpyHostToDevice));
  }
  float *q, *out;
  CK(hipMalloc(&q, n_heads * hd * 4));
  CK(hipMalloc(&out, n_heads * hd * 4));
  std::vector<float> hq(n_heads * hd);
  for (size_t i = 0; i < hq.size(); i++) hq[i] = 0.01f * (float)((int)(i % 37) - 18);
  CK(hipMemcpy(q, hq.data(), hq.size() * 4, hipMemcpyHostToDevice));
  std::vector<unsigned short> tab(65536);
  for (uint32_t x = 0; x < 65536; x++) {
    _Float16 hx;
    unsigned short b = (unsigned short)x;
    memcpy(&hx, &b, 2);
    _Float16 e = (_Float16)expf((float)hx);
    memcpy(&tab[x], &e, 2);
  }
  unsigned short* d_tab;
  CK(hipMalloc(&d_tab, 65536 * 2));
  CK(hipMemcpy(d_tab, tab.data(), 65536 * 2, hipMemcpyHostToDevice));
  signed char* xq;
  unsigned short* xd;
  int* xs;
  CK(hipMalloc(&xq, n_heads * hd));
  CK(hipMalloc(&xd, n_heads * hd / 32 * 2));
  CK(hipMalloc(&xs, n_heads * hd / 32 * 4));
  int* pos_d;
  CK(hipMalloc(&pos_d, 4));
  long long* stamps;
  CK(hipMalloc(&stamps, 256));
  char* wo;
  const size_t wo_bytes = (size_t)9437184 + 1048576;
  CK(hipMalloc(&wo, wo_bytes * 8));
  int* sink;
  CK(hipMalloc(&sink, 4));
  const size_t lds = (size_t)(seq_cap + hd) * 4;
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  for (int pos : {7, 23, 39, 71, 127, 220}) {
    CK(hipMemcpy(pos_d, &pos, 4, hipMemcpyHostToDevice));
    // phase stamps (no prefetch workgroups), a few layers' caches in turn; report the median launch
    std::vector<std::vector<long long>> runs;
    for (int it = 0; it < 9; it++) {
      const int l = it % L;
      k_attn<true, true><<<dim3(n_heads), 256, lds, st>>>(q, kc + l * kvb, vc + l * kvb, pos_d, d_tab, out, xq, xd, (void*)xs, n_heads, n_kv, hd,
                                                          seq_cap, PrefetchPlan{}, 0, stamps);
      CK(hipStreamSynchronize(st));
      std::vector<long long> s(6);
      CK(hipMemcpy(s.data(), stamps, 48, hipMemcpyDeviceToHost));
      runs.push_back(s);
    }
    std::sort(runs.begin(), runs.end(), [](const auto& a, const auto& b) { return a[5] - a[0] < b[5] - b[0]; });
    const auto& s = runs[runs.size() / 2];
    printf("pos %3d  workgroup 0 phases (shader cycles): q staged %5lld | scores %5lld | softmax %5lld | pv %5lld | quantize+store %5lld | total %5lld\n",
           pos, s[1] - s[0], s[2] - s[1], s[3] - s[2], s[4] - s[3], s[5] - s[4], s[5] - s[0]);
    {  // the LDS-staged kernel (k_attn_s), same stamps
      const int S = 224;
      const size_t lds_s = attn_s_lds_bytes(S, hd);
      CK(hipFuncSetAttribute((const void*)k_attn_s<128, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_s));
      CK(hipFuncSetAttribute((const void*)k_attn_s<128, false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_s));
      std::vector<std::vector<long long>> runs2;
      for (int it = 0; it < 9; it++) {
        const int l = it % L;
        k_attn_s<128, true><<<dim3(n_heads), 256, lds_s, st>>>(q, (const unsigned short*)(kc + l * kvb), (const unsigned short*)(vc + l * kvb), pos_d, d_tab,
                                                          out, xq, xd, (void*)xs, n_heads, n_kv, hd, seq_cap, S, PrefetchPlan{}, 0, stamps, AttnQ8K{});
        CK(hipStreamSynchronize(st));
        std::vector<long long> s2(10);
        CK(hipMemcpy(s2.data(), stamps, 80, hipMemcpyDeviceToHost));
        runs2.push_back(s2);
      }
      std::sort(runs2.begin(), runs2.end(), [](const auto& a, const auto& b) { return a[5] - a[0] < b[5] - b[0]; });
      const auto& s2 = runs2[runs2.size() / 2];
      printf("pos %3d  STAGED  workgroup 0 phases: staged %5lld | scores %5lld | softmax %5lld | pv %5lld | quantize+store %5lld | total %5lld\n", pos,
             s2[1] - s2[0], s2[2] - s2[1], s2[3] - s2[2], s2[4] - s2[3], s2[5] - s2[4], s2[5] - s2[0]);
      float best = 1e9f;
      for (int rep = 0; rep < 3; rep++) {
        CK(hipEventRecord(e0, st));
        for (int it = 0; it < 96; it++) {
          const int l = it % L;
          k_attn_s<128, false><<<dim3(n_heads), 256, lds_s, st>>>(q, (const unsigned short*)(kc + l * kvb), (const unsigned short*)(vc + l * kvb), pos_d, d_tab,
                                                             out, xq, xd, (void*)xs, n_heads, n_kv, hd, seq_cap, S, PrefetchPlan{}, 0, nullptr, AttnQ8K{});
        }
        CK(hipEventRecord(e1, st));
        CK(hipEventSynchronize(e1));
        float ms;
        CK(hipEventElapsedTime(&ms, e0, e1));
        best = std::min(best, ms * 1e3f / 96);
      }
      printf("pos %3d  STAGED  inside `staged`: loads issued %5lld | position known %5lld | q staged %5lld | K,V in LDS %5lld | barrier %5lld\n", pos,
             s2[6] - s2[0], s2[7] - s2[6], s2[8] - s2[7], s2[9] - s2[8], s2[1] - s2[9]);
      printf("pos %3d  launch STAGED 32 workgroups            : %.2f us\n", pos, best);
    }
    // launch time: back-to-back launches over different layers' caches, with and without the prefetch workgroups
    for (int spare : {0, 224}) {
      float best = 1e9f;
      for (int rep = 0; rep < 3; rep++) {
        CK(hipEventRecord(e0, st));
        for (int it = 0; it < 96; it++) {
          const int l = it % L;
          PrefetchPlan pf{};
          if (spare) {
            pf.p[0] = wo + (size_t)(it % 8) * wo_bytes;
            pf.n[0] = 9437184;
            pf.sink = sink;
          }
          k_attn<true, false><<<dim3(n_heads + spare), 256, lds, st>>>(q, kc + l * kvb, vc + l * kvb, pos_d, d_tab, out, xq, xd, (void*)xs, n_heads, n_kv,
                                                                       hd, seq_cap, pf, 0, nullptr);
        }
        CK(hipEventRecord(e1, st));
        CK(hipEventSynchronize(e1));
        float ms;
        CK(hipEventElapsedTime(&ms, e0, e1));
        best = std::min(best, ms * 1e3f / 96);
      }
      printf("pos %3d  launch %s: %.2f us\n", pos, spare ? "+ 224 prefetch workgroups (9.4 MB of wo)" : "32 workgroups only                      ", best);
    }
  }
  return 0;
}

// launch_lab.hip -- how much does one dependent kernel boundary cost on MI355X, as a function of what the
// kernel's critical path looks like?  (design input for the fused decode step: ~7 dependent stages/layer)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1);} } while (0)

__global__ void k_empty() {}
__global__ void k_inc(int* x) { if (threadIdx.x == 0 && blockIdx.x == 0) x[0] += 1; }                    // 1 load + store
__global__ void k_chain2(int* idx, int* data) { if (threadIdx.x == 0 && blockIdx.x == 0) { int p = idx[0]; data[p & 1023] += 1; idx[0] = p + 1; } }  // 2 dependent loads
__global__ void k_chain3(int* idx, int* tab, int* data) { if (threadIdx.x == 0 && blockIdx.x == 0) { int p = idx[0]; int q = tab[p & 1023]; data[q & 1023] += 1; idx[0] = p + 1; } }
// every WG of a wide grid touches memory written by the previous kernel (like a GEMV reading xq)
__global__ void k_wide(const int* __restrict__ in, int* __restrict__ out, int n) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) out[i] = in[(i * 7) % n] + 1;
}

int main() {
  hipStream_t st; CK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
  int *a, *b, *c; CK(hipMalloc(&a, 1 << 20)); CK(hipMalloc(&b, 1 << 20)); CK(hipMalloc(&c, 1 << 20));
  CK(hipMemset(a, 0, 1 << 20)); CK(hipMemset(b, 0, 1 << 20)); CK(hipMemset(c, 0, 1 << 20));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  const int N = 2000;
  auto timeit = [&](const char* name, auto launch) {
    for (int mode = 0; mode < 2; mode++) {
      hipGraph_t g = nullptr; hipGraphExec_t ex = nullptr;
      if (mode == 1) {
        CK(hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal));
        for (int i = 0; i < N; i++) launch(i);
        CK(hipStreamEndCapture(st, &g)); CK(hipGraphInstantiate(&ex, g, nullptr, nullptr, 0));
      }
      float best = 1e9;
      for (int rep = 0; rep < 3; rep++) {
        CK(hipEventRecord(e0, st));
        if (mode == 0) for (int i = 0; i < N; i++) launch(i); else CK(hipGraphLaunch(ex, st));
        CK(hipEventRecord(e1, st)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1)); if (ms < best) best = ms;
      }
      printf("%-34s %-6s %7.2f us per kernel\n", name, mode ? "graph" : "eager", best * 1000 / N);
      if (ex) { CK(hipGraphExecDestroy(ex)); CK(hipGraphDestroy(g)); }
    }
  };
  timeit("empty<<<1,64>>>", [&](int) { k_empty<<<1, 64, 0, st>>>(); });
  timeit("empty<<<2048,128>>>", [&](int) { k_empty<<<2048, 128, 0, st>>>(); });
  timeit("inc (1 dependent load)", [&](int) { k_inc<<<1, 64, 0, st>>>(a); });
  timeit("chain2 (2 dependent loads)", [&](int) { k_chain2<<<1, 64, 0, st>>>(a, b); });
  timeit("chain3 (3 dependent loads)", [&](int) { k_chain3<<<1, 64, 0, st>>>(a, b, c); });
  timeit("wide 2048x128 ping-pong", [&](int i) { if (i & 1) k_wide<<<2048, 128, 0, st>>>(a, b, 1 << 18); else k_wide<<<2048, 128, 0, st>>>(b, a, 1 << 18); });
  timeit("wide 256x1024 ping-pong", [&](int i) { if (i & 1) k_wide<<<256, 1024, 0, st>>>(a, b, 1 << 18); else k_wide<<<256, 1024, 0, st>>>(b, a, 1 << 18); });
  return 0;
}

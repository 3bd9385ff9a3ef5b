// upload_lab.hip -- how should a multi-GB mmap'd (pageable) weight file reach HBM?  (design input for
// crabml_hip_buf_from_cpu)  build: hipcc --offload-arch=gfx950 -O2 tools/upload_lab.hip -o /tmp/upload_lab
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1);} } while (0)
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main() {
  const size_t N = (size_t)2 << 30;  // 2 GiB
  char* src = (char*)malloc(N);
  for (size_t i = 0; i < N; i += 4096) src[i] = (char)i;  // touch pages
  void* dst; CK(hipMalloc(&dst, N));
  hipStream_t st; CK(hipStreamCreate(&st));
  CK(hipMemcpy(dst, src, 1 << 20, hipMemcpyHostToDevice));
  double t0 = now();
  CK(hipMemcpyAsync(dst, src, N, hipMemcpyHostToDevice, st)); CK(hipStreamSynchronize(st));
  double t1 = now();
  printf("pageable hipMemcpyAsync            %6.2f GB/s\n", N / (t1 - t0) / 1e9);
  t0 = now();
  for (size_t o = 0; o < N; o += (size_t)128 << 20) CK(hipMemcpyAsync((char*)dst + o, src + o, (size_t)128 << 20, hipMemcpyHostToDevice, st));
  CK(hipStreamSynchronize(st));
  t1 = now();
  printf("pageable, 128 MiB chunks           %6.2f GB/s\n", N / (t1 - t0) / 1e9);
  t0 = now();
  CK(hipHostRegister(src, N, hipHostRegisterDefault));
  double tr = now();
  CK(hipMemcpyAsync(dst, src, N, hipMemcpyHostToDevice, st)); CK(hipStreamSynchronize(st));
  t1 = now();
  CK(hipHostUnregister(src));
  double tu = now();
  printf("hipHostRegister %.3f s + copy %.3f s (%.1f GB/s) + unregister %.3f s => %6.2f GB/s overall\n", tr - t0, t1 - tr,
         N / (t1 - tr) / 1e9, tu - t1, N / (tu - t0) / 1e9);
  // pinned ring: 2 x 64 MiB, memcpy + async DMA
  const size_t C = (size_t)64 << 20;
  char* pin[2]; hipEvent_t ev[2];
  for (int i = 0; i < 2; i++) { CK(hipHostMalloc((void**)&pin[i], C, hipHostMallocDefault)); CK(hipEventCreate(&ev[i])); }
  t0 = now();
  int s = 0;
  for (size_t o = 0; o < N; o += C, s ^= 1) {
    CK(hipEventSynchronize(ev[s]));
    memcpy(pin[s], src + o, C);
    CK(hipMemcpyAsync((char*)dst + o, pin[s], C, hipMemcpyHostToDevice, st));
    CK(hipEventRecord(ev[s], st));
  }
  CK(hipStreamSynchronize(st));
  t1 = now();
  printf("memcpy -> pinned ring -> DMA       %6.2f GB/s\n", N / (t1 - t0) / 1e9);
  return 0;
}

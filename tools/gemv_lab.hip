// gemv_lab.hip -- design lab for the Q4_0 decode GEMV on MI355X (gfx950).
//
// Not part of the product: a standalone microbenchmark used to CHOOSE the weight layout and
// the wave/row mapping of crabml_amd/csrc/gemv.hip by measurement.  Each variant computes the
// same thing (W[m,k] Q4_0  x  activations pre-quantized to Q8_0) and is checked against a
// host loop before it is timed.  Timing cycles through enough distinct weight copies to defeat
// the 256 MiB Infinity Cache (each copy is touched once per sweep).
//
//   hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -o gemv_lab tools/gemv_lab.hip
//   ./gemv_lab            (prints one line per variant x shape)
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include <string>
#include <cmath>

#define CK(x)                                                                              \
  do {                                                                                     \
    hipError_t e_ = (x);                                                                   \
    if (e_ != hipSuccess) {                                                                \
      fprintf(stderr, "HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); \
      exit(1);                                                                             \
    }                                                                                      \
  } while (0)

typedef int i32x4 __attribute__((ext_vector_type(4)));
typedef int i32x2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}

__device__ __forceinline__ float h2f(unsigned short h) {
  _Float16 x;
  __builtin_memcpy(&x, &h, 2);
  return (float)x;
}

// integer part of one Q4_0 block against 32 int8 activations (xs = 8 dwords), minus 8*sum(x)
__device__ __forceinline__ int q4_0_block_dot(i32x4 q, i32x4 xlo, i32x4 xhi, int xsum) {
  int s = 0;
#pragma unroll
  for (int i = 0; i < 4; i++) {
    int w = q[i];
    int lo = w & 0x0F0F0F0F;
    int hi = (w >> 4) & 0x0F0F0F0F;
    s = __builtin_amdgcn_sdot4(lo, xlo[i], s, false);
    s = __builtin_amdgcn_sdot4(hi, xhi[i], s, false);
  }
  return s - 8 * xsum;
}

// ---------------------------------------------------------------------------------------------
// Variant P (planes): qs plane [m][nb] x 16 B, d plane [m][nb] f16.  One wave owns R rows at a
// time; lane l walks blocks l, l+64, ...; activations are re-read (L1/L2 hits) per row group.
// ---------------------------------------------------------------------------------------------
template <int R, bool NT>
__global__ __launch_bounds__(256) void k_planes(const i32x4* __restrict__ qs, const unsigned short* __restrict__ wd,
                                                const i32x4* __restrict__ xq, const unsigned short* __restrict__ xd,
                                                const int* __restrict__ xsum, float* __restrict__ out, int m, int nb) {
  const int lane = threadIdx.x & 63;
  const int wave = (blockIdx.x * (blockDim.x >> 6)) + (threadIdx.x >> 6);
  const int row0 = wave * R;
  if (row0 >= m) return;
  float acc[R];
#pragma unroll
  for (int r = 0; r < R; r++) acc[r] = 0.f;
  for (int b = lane; b < nb; b += 64) {
    i32x4 q[R];
    unsigned short dw[R];
#pragma unroll
    for (int r = 0; r < R; r++) {
      int row = row0 + r < m ? row0 + r : m - 1;
      size_t idx = (size_t)row * nb + b;
      q[r] = NT ? __builtin_nontemporal_load(qs + idx) : qs[idx];
      dw[r] = NT ? __builtin_nontemporal_load(wd + idx) : wd[idx];
    }
    i32x4 xlo = xq[2 * b], xhi = xq[2 * b + 1];
    float dx = h2f(xd[b]);
    int xs = xsum[b];
#pragma unroll
    for (int r = 0; r < R; r++) {
      int si = q4_0_block_dot(q[r], xlo, xhi, xs);
      acc[r] += ((float)si * h2f(dw[r])) * dx;
    }
  }
#pragma unroll
  for (int r = 0; r < R; r++) {
    float s = wave_sum(acc[r]);
    if (lane == 0 && row0 + r < m) out[row0 + r] = s;
  }
}

// ---------------------------------------------------------------------------------------------
// Variant H (hoisted): same planes, but the wave keeps ITS slice of the activations in
// registers (NCH = nb/64 chunks, compile time) and grid-strides over rows, U rows per step.
// ---------------------------------------------------------------------------------------------
template <int NCH, int U, bool NT>
__global__ __launch_bounds__(256) void k_hoist(const i32x4* __restrict__ qs, const unsigned short* __restrict__ wd,
                                               const i32x4* __restrict__ xq, const unsigned short* __restrict__ xd,
                                               const int* __restrict__ xsum, float* __restrict__ out, int m) {
  constexpr int nb = NCH * 64;
  const int lane = threadIdx.x & 63;
  const int wave = (blockIdx.x * (blockDim.x >> 6)) + (threadIdx.x >> 6);
  const int nwaves = gridDim.x * (blockDim.x >> 6);
  i32x4 xlo[NCH], xhi[NCH];
  float dx[NCH];
  int xs[NCH];
#pragma unroll
  for (int c = 0; c < NCH; c++) {
    int b = c * 64 + lane;
    xlo[c] = xq[2 * b];
    xhi[c] = xq[2 * b + 1];
    dx[c] = h2f(xd[b]);
    xs[c] = xsum[b];
  }
  for (int row0 = wave * U; row0 < m; row0 += nwaves * U) {
    float acc[U];
    i32x4 q[U][NCH];
    unsigned short dw[U][NCH];
#pragma unroll
    for (int u = 0; u < U; u++) {
      int row = row0 + u < m ? row0 + u : m - 1;
#pragma unroll
      for (int c = 0; c < NCH; c++) {
        size_t idx = (size_t)row * nb + c * 64 + lane;
        q[u][c] = NT ? __builtin_nontemporal_load(qs + idx) : qs[idx];
        dw[u][c] = NT ? __builtin_nontemporal_load(wd + idx) : wd[idx];
      }
    }
#pragma unroll
    for (int u = 0; u < U; u++) {
      acc[u] = 0.f;
#pragma unroll
      for (int c = 0; c < NCH; c++) {
        int si = q4_0_block_dot(q[u][c], xlo[c], xhi[c], xs[c]);
        acc[u] += ((float)si * h2f(dw[u][c])) * dx[c];
      }
    }
#pragma unroll
    for (int u = 0; u < U; u++) {
      float s = wave_sum(acc[u]);
      if (lane == 0 && row0 + u < m) out[row0 + u] = s;
    }
  }
}

// ---------------------------------------------------------------------------------------------
// Variant A (AoS): the GGUF byte layout untouched: 18-byte blocks at 2-byte alignment.
// Each lane reads its block with nine 2-byte loads (the only alignment the format guarantees).
// ---------------------------------------------------------------------------------------------
template <int R>
__global__ __launch_bounds__(256) void k_aos(const unsigned short* __restrict__ w, const i32x4* __restrict__ xq,
                                             const unsigned short* __restrict__ xd, const int* __restrict__ xsum,
                                             float* __restrict__ out, int m, int nb) {
  const int lane = threadIdx.x & 63;
  const int wave = (blockIdx.x * (blockDim.x >> 6)) + (threadIdx.x >> 6);
  const int row0 = wave * R;
  if (row0 >= m) return;
  float acc[R];
#pragma unroll
  for (int r = 0; r < R; r++) acc[r] = 0.f;
  for (int b = lane; b < nb; b += 64) {
    i32x4 xlo = xq[2 * b], xhi = xq[2 * b + 1];
    float dx = h2f(xd[b]);
    int xs = xsum[b];
#pragma unroll
    for (int r = 0; r < R; r++) {
      int row = row0 + r < m ? row0 + r : m - 1;
      const unsigned short* p = w + ((size_t)row * nb + b) * 9;
      unsigned short h[9];
#pragma unroll
      for (int i = 0; i < 9; i++) h[i] = p[i];
      i32x4 q;
#pragma unroll
      for (int i = 0; i < 4; i++) q[i] = (int)((unsigned)h[1 + 2 * i] | ((unsigned)h[2 + 2 * i] << 16));
      int si = q4_0_block_dot(q, xlo, xhi, xs);
      acc[r] += ((float)si * h2f(h[0])) * dx;
    }
  }
#pragma unroll
  for (int r = 0; r < R; r++) {
    float s = wave_sum(acc[r]);
    if (lane == 0 && row0 + r < m) out[row0 + r] = s;
  }
}

// ---------------------------------------------------------------------------------------------
// Variant T (tiles): [row][chunk] tiles of 64 blocks: 1024 B of qs followed by 128 B of d, so a
// wave step reads ONE contiguous 1152-byte segment.  Needs nb % 64 == 0.
// ---------------------------------------------------------------------------------------------
template <int R, bool NT>
__global__ __launch_bounds__(256) void k_tiles(const unsigned char* __restrict__ w, const i32x4* __restrict__ xq,
                                               const unsigned short* __restrict__ xd, const int* __restrict__ xsum,
                                               float* __restrict__ out, int m, int nch) {
  const int lane = threadIdx.x & 63;
  const int wave = (blockIdx.x * (blockDim.x >> 6)) + (threadIdx.x >> 6);
  const int row0 = wave * R;
  if (row0 >= m) return;
  float acc[R];
#pragma unroll
  for (int r = 0; r < R; r++) acc[r] = 0.f;
  for (int c = 0; c < nch; c++) {
    i32x4 q[R];
    unsigned short dw[R];
#pragma unroll
    for (int r = 0; r < R; r++) {
      int row = row0 + r < m ? row0 + r : m - 1;
      const unsigned char* t = w + ((size_t)row * nch + c) * 1152;
      const i32x4* tq = (const i32x4*)t + lane;
      const unsigned short* td = (const unsigned short*)(t + 1024) + lane;
      q[r] = NT ? __builtin_nontemporal_load(tq) : *tq;
      dw[r] = NT ? __builtin_nontemporal_load(td) : *td;
    }
    int b = c * 64 + lane;
    i32x4 xlo = xq[2 * b], xhi = xq[2 * b + 1];
    float dx = h2f(xd[b]);
    int xs = xsum[b];
#pragma unroll
    for (int r = 0; r < R; r++) {
      int si = q4_0_block_dot(q[r], xlo, xhi, xs);
      acc[r] += ((float)si * h2f(dw[r])) * dx;
    }
  }
#pragma unroll
  for (int r = 0; r < R; r++) {
    float s = wave_sum(acc[r]);
    if (lane == 0 && row0 + r < m) out[row0 + r] = s;
  }
}

// pure streaming read (the box's achievable HBM read ceiling for 16-byte loads)
template <bool NT>
__global__ __launch_bounds__(256) void k_stream(const i32x4* __restrict__ p, size_t n16, int* __restrict__ sink) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  size_t stride = (size_t)gridDim.x * blockDim.x;
  int acc = 0;
  for (; i + 3 * stride < n16; i += 4 * stride) {
    i32x4 a = NT ? __builtin_nontemporal_load(p + i) : p[i];
    i32x4 b = NT ? __builtin_nontemporal_load(p + i + stride) : p[i + stride];
    i32x4 c = NT ? __builtin_nontemporal_load(p + i + 2 * stride) : p[i + 2 * stride];
    i32x4 d = NT ? __builtin_nontemporal_load(p + i + 3 * stride) : p[i + 3 * stride];
    acc += a.x ^ a.y ^ a.z ^ a.w ^ b.x ^ b.y ^ b.z ^ b.w ^ c.x ^ c.y ^ c.z ^ c.w ^ d.x ^ d.y ^ d.z ^ d.w;
  }
  for (; i < n16; i += stride) {
    i32x4 a = p[i];
    acc += a.x ^ a.y ^ a.z ^ a.w;
  }
  if (acc == 0x7fffffff) sink[0] = acc;
}

// ---------------------------------------------------------------------------------------------
static uint32_t rng_state = 12345u;
static inline uint32_t rnd() {
  rng_state ^= rng_state << 13;
  rng_state ^= rng_state >> 17;
  rng_state ^= rng_state << 5;
  return rng_state;
}
static uint16_t f2h(float f) {
  _Float16 h = (_Float16)f;
  uint16_t u;
  memcpy(&u, &h, 2);
  return u;
}
static float h2f_host(uint16_t u) {
  _Float16 h;
  memcpy(&h, &u, 2);
  return (float)h;
}

struct Shape {
  int m, k;
};

int main(int argc, char** argv) {
  int reps = argc > 1 ? atoi(argv[1]) : 3;
  hipDeviceProp_t prop;
  CK(hipGetDeviceProperties(&prop, 0));
  printf("# device %s CUs=%d clock=%d MHz\n", prop.name, prop.multiProcessorCount, prop.clockRate / 1000);
  hipStream_t st;
  CK(hipStreamCreate(&st));
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));

  {  // streaming ceiling
    size_t bytes = (size_t)2 << 30;
    void* p;
    CK(hipMalloc(&p, bytes));
    CK(hipMemset(p, 1, bytes));
    int* sink;
    CK(hipMalloc(&sink, 4));
    for (int nt = 0; nt < 2; nt++)
      for (int grid : {1024, 2048, 4096, 8192}) {
        for (int it = 0; it < 2; it++) {
          CK(hipEventRecord(e0, st));
          if (nt)
            k_stream<true><<<grid, 256, 0, st>>>((const i32x4*)p, bytes / 16, sink);
          else
            k_stream<false><<<grid, 256, 0, st>>>((const i32x4*)p, bytes / 16, sink);
          CK(hipEventRecord(e1, st));
          CK(hipEventSynchronize(e1));
          float ms;
          CK(hipEventElapsedTime(&ms, e0, e1));
          if (it == 1) printf("stream_read nt=%d grid=%d  %.1f GB/s\n", nt, grid, bytes / ms / 1e6);
        }
      }
    CK(hipFree(p));
    CK(hipFree(sink));
  }

  std::vector<Shape> shapes = {{4096, 4096}, {1024, 4096}, {14336, 4096}, {4096, 14336}, {128256, 4096}};
  for (auto sh : shapes) {
    const int m = sh.m, k = sh.k, nb = k / 32, nch = nb / 64;
    const size_t nblk = (size_t)m * nb;
    const size_t wbytes = nblk * 18;
    int copies = (int)((600ull << 20) / wbytes) + 1;
    if (copies > 256) copies = 256;
    if (copies < 2) copies = 2;
    // host data: one base matrix, copies differ by a cheap xor so contents are not identical
    std::vector<uint8_t> aos(wbytes);
    for (size_t i = 0; i < nblk; i++) {
      float d = 0.002f + (rnd() % 1000) * 1.8e-5f;
      uint16_t h = f2h(d);
      memcpy(&aos[i * 18], &h, 2);
      for (int j = 0; j < 16; j += 4) {
        uint32_t r = rnd();
        memcpy(&aos[i * 18 + 2 + j], &r, 4);
      }
    }
    std::vector<int8_t> xq(k);
    std::vector<uint16_t> xd(nb);
    std::vector<int> xsum(nb);
    for (int b = 0; b < nb; b++) {
      int s = 0;
      for (int j = 0; j < 32; j++) {
        int v = (int)(rnd() % 255) - 127;
        xq[b * 32 + j] = (int8_t)v;
        s += v;
      }
      xsum[b] = s;
      xd[b] = f2h(0.01f + (rnd() % 100) * 1e-4f);
    }
    // host reference for the first 64 rows + last row of copy 0
    std::vector<int> check_rows;
    for (int r = 0; r < 64 && r < m; r++) check_rows.push_back(r);
    check_rows.push_back(m - 1);
    std::vector<double> ref(check_rows.size());
    for (size_t ci = 0; ci < check_rows.size(); ci++) {
      int r = check_rows[ci];
      double acc = 0;
      for (int b = 0; b < nb; b++) {
        const uint8_t* blk = &aos[((size_t)r * nb + b) * 18];
        uint16_t h;
        memcpy(&h, blk, 2);
        int si = 0;
        for (int j = 0; j < 16; j++) {
          si += ((blk[2 + j] & 0xF) - 8) * xq[b * 32 + j] + ((blk[2 + j] >> 4) - 8) * xq[b * 32 + 16 + j];
        }
        acc += (double)si * h2f_host(h) * h2f_host(xd[b]);
      }
      ref[ci] = acc;
    }
    // layouts
    std::vector<uint8_t> planes(wbytes), tiles;
    for (size_t i = 0; i < nblk; i++) {
      memcpy(&planes[i * 16], &aos[i * 18 + 2], 16);
      memcpy(&planes[nblk * 16 + i * 2], &aos[i * 18], 2);
    }
    bool can_tile = (nb % 64 == 0);
    if (can_tile) {
      tiles.resize(wbytes);
      for (int r = 0; r < m; r++)
        for (int c = 0; c < nch; c++) {
          uint8_t* t = &tiles[((size_t)r * nch + c) * 1152];
          for (int l = 0; l < 64; l++) {
            size_t i = (size_t)r * nb + c * 64 + l;
            memcpy(t + l * 16, &aos[i * 18 + 2], 16);
            memcpy(t + 1024 + l * 2, &aos[i * 18], 2);
          }
        }
    }
    uint8_t* d_w;  // copies of one layout at a time
    CK(hipMalloc((void**)&d_w, wbytes * copies));
    int8_t* d_xq;
    uint16_t* d_xd;
    int* d_xs;
    float* d_out;
    CK(hipMalloc((void**)&d_xq, k));
    CK(hipMalloc((void**)&d_xd, nb * 2));
    CK(hipMalloc((void**)&d_xs, nb * 4));
    CK(hipMalloc((void**)&d_out, (size_t)m * 4 * copies));
    CK(hipMemcpy(d_xq, xq.data(), k, hipMemcpyHostToDevice));
    CK(hipMemcpy(d_xd, xd.data(), nb * 2, hipMemcpyHostToDevice));
    CK(hipMemcpy(d_xs, xsum.data(), nb * 4, hipMemcpyHostToDevice));

    auto upload = [&](const std::vector<uint8_t>& src) {
      for (int c = 0; c < copies; c++) CK(hipMemcpy(d_w + (size_t)c * wbytes, src.data(), wbytes, hipMemcpyHostToDevice));
    };
    std::vector<float> hout(m);
    auto run_variant = [&](const char* name, auto launch) {
      CK(hipMemset(d_out, 0, (size_t)m * 4));
      launch(0);
      CK(hipStreamSynchronize(st));
      CK(hipMemcpy(hout.data(), d_out, (size_t)m * 4, hipMemcpyDeviceToHost));
      double maxrel = 0;
      for (size_t ci = 0; ci < check_rows.size(); ci++) {
        double g = hout[check_rows[ci]], r = ref[ci];
        double rel = fabs(g - r) / (fabs(r) + 1e-3);
        if (rel > maxrel) maxrel = rel;
      }
      float best = 1e30f;
      for (int rep = 0; rep < reps; rep++) {
        CK(hipEventRecord(e0, st));
        for (int c = 0; c < copies; c++) launch(c);
        CK(hipEventRecord(e1, st));
        CK(hipEventSynchronize(e1));
        float ms;
        CK(hipEventElapsedTime(&ms, e0, e1));
        if (ms < best) best = ms;
      }
      double us = best * 1000.0 / copies;
      double gbs = (double)(wbytes + (size_t)k * 4 + (size_t)m * 4) / us / 1e3;
      printf("%-28s m=%-6d k=%-5d  %8.2f us  %8.1f GB/s  %s(maxrel %.1e)\n", name, m, k, us, gbs,
             maxrel < 1e-4 ? "ok " : "WRONG ", maxrel);
      fflush(stdout);
    };

    // ---- planes
    upload(planes);
#define PL(R, NT, TPB)                                                                                         \
  run_variant("planes R=" #R " nt=" #NT " tpb=" #TPB, [&](int c) {                                             \
    const uint8_t* base = d_w + (size_t)c * wbytes;                                                            \
    int waves = (m + R - 1) / R;                                                                               \
    int wpb = TPB / 64;                                                                                        \
    int grid = (waves + wpb - 1) / wpb;                                                                        \
    k_planes<R, NT><<<grid, TPB, 0, st>>>((const i32x4*)base, (const unsigned short*)(base + nblk * 16),       \
                                          (const i32x4*)d_xq, d_xd, d_xs, d_out + (size_t)c * m, m, nb);       \
  })
    PL(1, false, 256);
    PL(1, true, 256);
    PL(2, false, 256);
    PL(2, true, 256);
    PL(4, true, 256);
    PL(4, false, 256);
    PL(8, true, 256);
    PL(2, true, 64);
    PL(2, true, 128);
    PL(4, true, 64);
    PL(2, true, 512);
#define HO(NCH, U, NT, WPC)                                                                                    \
  if (nch == NCH && nb % 64 == 0)                                                                              \
  run_variant("hoist U=" #U " nt=" #NT " waves/CU=" #WPC, [&](int c) {                                         \
    const uint8_t* base = d_w + (size_t)c * wbytes;                                                            \
    int grid = prop.multiProcessorCount * WPC / 4;                                                             \
    k_hoist<NCH, U, NT><<<grid, 256, 0, st>>>((const i32x4*)base, (const unsigned short*)(base + nblk * 16),   \
                                              (const i32x4*)d_xq, d_xd, d_xs, d_out + (size_t)c * m, m);       \
  })
    HO(2, 1, true, 16);
    HO(2, 2, true, 16);
    HO(2, 4, true, 16);
    HO(2, 2, true, 8);
    HO(2, 4, true, 8);
    HO(2, 2, true, 32);
    HO(2, 2, false, 16);
    HO(7, 1, true, 16);
    HO(7, 2, true, 16);
    HO(7, 1, true, 8);
    HO(7, 2, true, 8);
    HO(7, 1, false, 16);
    // ---- tiles
    if (can_tile) {
      upload(tiles);
#define TI(R, NT)                                                                                              \
  run_variant("tiles R=" #R " nt=" #NT, [&](int c) {                                                           \
    const uint8_t* base = d_w + (size_t)c * wbytes;                                                            \
    int waves = (m + R - 1) / R;                                                                               \
    int grid = (waves + 3) / 4;                                                                                \
    k_tiles<R, NT><<<grid, 256, 0, st>>>(base, (const i32x4*)d_xq, d_xd, d_xs, d_out + (size_t)c * m, m, nch); \
  })
      TI(1, true);
      TI(2, true);
      TI(4, true);
      TI(2, false);
    }
    // ---- AoS
    upload(aos);
#define AO(R)                                                                                                  \
  run_variant("aos R=" #R, [&](int c) {                                                                        \
    const uint8_t* base = d_w + (size_t)c * wbytes;                                                            \
    int waves = (m + R - 1) / R;                                                                               \
    int grid = (waves + 3) / 4;                                                                                \
    k_aos<R><<<grid, 256, 0, st>>>((const unsigned short*)base, (const i32x4*)d_xq, d_xd, d_xs,                \
                                   d_out + (size_t)c * m, m, nb);                                              \
  })
    AO(1);
    AO(2);
    CK(hipFree(d_w));
    CK(hipFree(d_xq));
    CK(hipFree(d_xd));
    CK(hipFree(d_xs));
    CK(hipFree(d_out));
  }
  return 0;
}

// elementwise.hip -- the small ops around the GEMV, one launch per crabml `Tensor` method.
//
// Every kernel reproduces the reference CPU arithmetic bit-for-bit (same rounding points, same
// summation order where the reference's order is sequential), so that the only fp divergence of the
// whole decode step is the GEMV's block-term summation order:
//   rms_norm   rms_norm.rs:9-47   (32-lane ordered chunk sums, chunk sums added serially, true division)
//   rope       rope.rs:10-80      (cos/sin of the iterated-theta recurrence come from the host libm)
//   softmax    softmax.rs:11-57   (max, exp via the f16->f16 table, sequential sum, division)
//   silu/gelu  silu.rs:6-13, gelu.rs:11-17 (f16 table lookups)
//   add/mul    arithmetic.rs:5-68 (cyclic broadcast of rhs)
//   contiguous contiguous.rs:6-66, concatenate concatenate.rs:12-204 (f32->f16 = RNE)
//   dequantize rows: BlockQ*::dequantize (buf_q8_0.rs:18-23, buf_q4_0.rs:18-27, buf_q4_1.rs:19-30,
//              buf_q4_k.rs:24-47, buf_q8_k.rs:15-20)
// These are µs-scale, launch-bound kernels; the fused decode path (fused.hip) folds them into the
// GEMV producers/consumers.
#include "dequant.hpp"
#include "kernels.hpp"

namespace crabml_hip {

__global__ __launch_bounds__(256) void k_binary(int op, float* __restrict__ a, size_t na, const float* __restrict__ b,
                                                size_t nb) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= na) return;
  float bv = b[nb == 1 ? 0 : i % nb];
  a[i] = op == 0 ? a[i] + bv : a[i] * bv;
}
void launch_binary(hipStream_t st, int op, float* a, size_t na, const float* b, size_t nb) {
  if (na == 0) return;
  k_binary<<<(unsigned)((na + 255) / 256), 256, 0, st>>>(op, a, na, b, nb);
}

__global__ __launch_bounds__(256) void k_scale(float* __restrict__ a, size_t n, float f) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) a[i] = a[i] * f;
}
void launch_scale(hipStream_t st, float* a, size_t n, float f) {
  if (n == 0) return;
  k_scale<<<(unsigned)((n + 255) / 256), 256, 0, st>>>(a, n, f);
}

// exp_f32_cached (buf_f32.rs:29-35): input rounded to f16, output is the f16 table entry
__device__ __forceinline__ float exp_cached(float x, const unsigned short* __restrict__ table) {
  return h2f(table[f2h(x)]);
}

__global__ __launch_bounds__(256) void k_silu(float* __restrict__ x, size_t n, const unsigned short* __restrict__ tab) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  float v = x[i];
  float nexp = exp_cached(-v, tab);
  x[i] = v / (1.0f + nexp);
}
void launch_silu(hipStream_t st, float* x, size_t n, const uint16_t* tab) {
  if (n == 0) return;
  k_silu<<<(unsigned)((n + 255) / 256), 256, 0, st>>>(x, n, tab);
}

__global__ __launch_bounds__(256) void k_gelu(float* __restrict__ x, size_t n, const unsigned short* __restrict__ tab) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) x[i] = h2f(tab[f2h(x[i])]);
}
void launch_gelu(hipStream_t st, float* x, size_t n, const uint16_t* tab) {
  if (n == 0) return;
  k_gelu<<<(unsigned)((n + 255) / 256), 256, 0, st>>>(x, n, tab);
}

// ---- rms_norm: one workgroup per row ---------------------------------------------------------------
// thread t owns the 32-element chunks t, t+256, ...: each chunk sum is the ordered left-to-right sum of
// squares starting from -0.0 (std::simd reduce_sum = simd_reduce_add_ordered), chunk sums are then added
// in chunk order by one thread -- the exact association of rms_norm.rs:35-40.
__global__ __launch_bounds__(256) void k_rms_norm(float* __restrict__ x, size_t cols, float eps) {
  extern __shared__ float chunk_sums[];
  __shared__ float s_rms;
  float* v = x + (size_t)blockIdx.x * cols;
  const size_t nchunks = cols / 32;
  for (size_t c = threadIdx.x; c < nchunks; c += blockDim.x) {
    const f32x4* p = (const f32x4*)(v + c * 32);
    float s = -0.0f;
#pragma unroll
    for (int j = 0; j < 8; j++) {
      f32x4 t = p[j];
      s += t[0] * t[0];
      s += t[1] * t[1];
      s += t[2] * t[2];
      s += t[3] * t[3];
    }
    chunk_sums[c] = s;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    float sum = 0.0f;
    for (size_t c = 0; c < nchunks; c++) sum += chunk_sums[c];
    s_rms = sqrtf(sum / (float)cols + eps);
  }
  __syncthreads();
  const float rms = s_rms;
  for (size_t i = threadIdx.x; i < nchunks * 32; i += blockDim.x) v[i] = v[i] / rms;
}
void launch_rms_norm(hipStream_t st, float* x, size_t rows, size_t cols, float eps) {
  if (rows == 0 || cols == 0) return;
  size_t lds = (cols / 32) * sizeof(float);
  k_rms_norm<<<(unsigned)rows, 256, lds, st>>>(x, cols, eps);
}

// ---- softmax: one workgroup per row ----------------------------------------------------------------
__global__ __launch_bounds__(256) void k_softmax(float* __restrict__ x, size_t cols,
                                                 const unsigned short* __restrict__ tab) {
  __shared__ float s_part[4];
  __shared__ float s_bcast;
  float* v = x + (size_t)blockIdx.x * cols;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  float mx = -INFINITY;
  for (size_t i = threadIdx.x; i < cols; i += blockDim.x) mx = fmaxf(mx, v[i]);
  mx = wave_max_f32(mx);
  if (lane == 0) s_part[wave] = mx;
  __syncthreads();
  mx = fmaxf(fmaxf(s_part[0], s_part[1]), fmaxf(s_part[2], s_part[3]));
  for (size_t i = threadIdx.x; i < cols; i += blockDim.x) v[i] = exp_cached(v[i] - mx, tab);
  __syncthreads();
  if (threadIdx.x == 0) {  // the reference accumulates the row sum sequentially (softmax.rs:43-48)
    float sum = 0.0f;
    for (size_t i = 0; i < cols; i++) sum += v[i];
    s_bcast = sum;
  }
  __syncthreads();
  const float sum = s_bcast;
  for (size_t i = threadIdx.x; i < cols; i += blockDim.x) v[i] = v[i] / sum;
}
void launch_softmax(hipStream_t st, float* x, size_t rows, size_t cols, const uint16_t* tab) {
  if (rows == 0 || cols == 0) return;
  k_softmax<<<(unsigned)rows, 256, 0, st>>>(x, cols, tab);
}

// ---- rope ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_rope(float* __restrict__ x, size_t n_heads, size_t head_dim, int mode,
                                              size_t npairs, RopeTable tab) {
  size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= n_heads * npairs) return;
  size_t h = t / npairs, i = t % npairs;
  float c = tab.cs[2 * i], s = tab.cs[2 * i + 1];
  float* chunk = x + h * head_dim;
  size_t i0 = mode == 0 ? 2 * i : i;
  size_t i1 = mode == 0 ? 2 * i + 1 : i + head_dim / 2;
  float qp0 = chunk[i0], qp1 = chunk[i1];
  chunk[i0] = qp0 * c - qp1 * s;
  chunk[i1] = qp0 * s + qp1 * c;
}
void launch_rope(hipStream_t st, float* x, size_t n_heads, size_t head_dim, int mode, size_t rope_dims,
                 const RopeTable& tab) {
  // Llama mode steps i by 2 over [0, rope_dims): ceil(rope_dims/2) pairs; Neox: rope_dims/2 pairs
  size_t npairs = mode == 0 ? (rope_dims + 1) / 2 : rope_dims / 2;
  size_t total = n_heads * npairs;
  if (total == 0) return;
  k_rope<<<(unsigned)((total + 255) / 256), 256, 0, st>>>(x, n_heads, head_dim, mode, npairs, tab);
}

// ---- contiguous / concatenate: strided copies ---------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(256) void k_contiguous(const T* __restrict__ src, T* __restrict__ dst, size_t s0,
                                                    size_t s1, size_t s2, size_t st0, size_t st1, size_t st2) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  size_t total = s0 * s1 * s2;
  if (i >= total) return;
  size_t z = i % s2, y = (i / s2) % s1, x = i / (s1 * s2);
  dst[i] = src[x * st0 + y * st1 + z * st2];
}
void launch_contiguous(hipStream_t st, const void* src, void* dst, int elem_size, const size_t shape[3],
                       const size_t strides[3]) {
  size_t total = shape[0] * shape[1] * shape[2];
  if (total == 0) return;
  unsigned grid = (unsigned)((total + 255) / 256);
  if (elem_size == 4)
    k_contiguous<float><<<grid, 256, 0, st>>>((const float*)src, (float*)dst, shape[0], shape[1], shape[2], strides[0],
                                              strides[1], strides[2]);
  else
    k_contiguous<unsigned short><<<grid, 256, 0, st>>>((const unsigned short*)src, (unsigned short*)dst, shape[0],
                                                       shape[1], shape[2], strides[0], strides[1], strides[2]);
}

template <typename D, typename S>
__global__ __launch_bounds__(256) void k_concat(D* __restrict__ dst, size_t dst_off, size_t d0, size_t d1, size_t d2,
                                                const S* __restrict__ src, size_t s0, size_t s1, size_t s2, size_t t0,
                                                size_t t1, size_t t2) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  size_t total = s0 * s1 * s2;
  if (i >= total) return;
  size_t z = i % s2, y = (i / s2) % s1, x = i / (s1 * s2);
  S v = src[x * t0 + y * t1 + z * t2];
  size_t o = dst_off + x * d0 + y * d1 + z * d2;
  if constexpr (sizeof(D) == 2 && sizeof(S) == 4)
    dst[o] = f2h(v);  // vec_convert_f16_f32: RNE (buf_f16.rs:165-173)
  else
    dst[o] = v;
}
void launch_concatenate(hipStream_t st, void* dst, int dst_f16, size_t dst_off, const size_t ds[3], const void* src,
                        int src_f16, const size_t shape[3], const size_t ss[3]) {
  size_t total = shape[0] * shape[1] * shape[2];
  if (total == 0) return;
  unsigned grid = (unsigned)((total + 255) / 256);
  if (!dst_f16 && !src_f16)
    k_concat<float, float><<<grid, 256, 0, st>>>((float*)dst, dst_off, ds[0], ds[1], ds[2], (const float*)src, shape[0],
                                                 shape[1], shape[2], ss[0], ss[1], ss[2]);
  else if (dst_f16 && src_f16)
    k_concat<unsigned short, unsigned short><<<grid, 256, 0, st>>>((unsigned short*)dst, dst_off, ds[0], ds[1], ds[2],
                                                                   (const unsigned short*)src, shape[0], shape[1],
                                                                   shape[2], ss[0], ss[1], ss[2]);
  else
    k_concat<unsigned short, float><<<grid, 256, 0, st>>>((unsigned short*)dst, dst_off, ds[0], ds[1], ds[2],
                                                          (const float*)src, shape[0], shape[1], shape[2], ss[0], ss[1],
                                                          ss[2]);
}

// ---- dequantize a run of elements (embedding lookup) ---------------------------------------------------
// One thread per output element; dequant_elem() (dequant.hpp) evaluates the reference's expression.
__global__ __launch_bounds__(256) void k_dequant(const char* __restrict__ w, int dtype, size_t off_scale,
                                                 size_t start, size_t n, void* __restrict__ dst, int dst_f16) {
  size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= n) return;
  float v = dequant_elem(w, dtype, off_scale, start + t);
  if (dst_f16)
    ((unsigned short*)dst)[t] = f2h(v);
  else
    ((float*)dst)[t] = v;
}
void launch_dequant_row(hipStream_t st, const crabml_hip_buf* src, size_t start, size_t n, void* dst, int dst_f16) {
  if (n == 0) return;
  k_dequant<<<(unsigned)((n + 255) / 256), 256, 0, st>>>((const char*)src->ptr, (int)src->dtype, src->wl.off_scale,
                                                         start, n, dst, dst_f16);
}

// ---- weight upload: GGUF blocks -> planes -------------------------------------------------------------------
// One thread per 2-byte unit (every block size, piece offset and piece length of the supported formats is even).
// Pure byte moves; runs once per tensor behind the host-to-device copy of the raw bytes.
__global__ __launch_bounds__(256) void k_repack(const unsigned short* __restrict__ raw, unsigned short* __restrict__ base,
                                                size_t blk0, size_t n_blocks, int bb2, RepackPlan plan) {
  const size_t u = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (u >= n_blocks * (size_t)bb2) return;
  const size_t b = u / bb2;
  const int o = (int)(u - b * bb2);
  const unsigned short v = raw[u];
  for (int s = 0; s < plan.nseg; s++) {
    const int r = o - plan.src_off2[s];
    if (r >= 0 && r < plan.len2[s]) {
      base[plan.dst_off[s] / 2 + (blk0 + b) * plan.stride2[s] + r] = v;
      return;
    }
  }
}

void launch_repack(hipStream_t st, const void* raw, void* base, size_t blk0, size_t n_blocks, int bb, const RepackPlan& plan) {
  if (n_blocks == 0) return;
  const size_t units = n_blocks * (size_t)(bb / 2);
  k_repack<<<(unsigned)((units + 255) / 256), 256, 0, st>>>((const unsigned short*)raw, (unsigned short*)base, blk0, n_blocks,
                                                            bb / 2, plan);
}

// Q4_K: scales[12] (8 x (6-bit scale, 6-bit min), util.rs:19-27) -> pair-major 24-bit fields (common.hpp), in place
__global__ __launch_bounds__(256) void k_q4k_pack_scales(unsigned char* __restrict__ hdr, size_t blk0, size_t n_blocks) {
  const size_t b = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= n_blocks) return;
  unsigned char* q = hdr + (blk0 + b) * 16 + 4;
  unsigned sc[8], mn[8];
  for (int j = 0; j < 8; j++) {
    if (j < 4) {
      sc[j] = q[j] & 63u;
      mn[j] = q[j + 4] & 63u;
    } else {
      sc[j] = (q[j + 4] & 0xFu) | ((unsigned)(q[j - 4] >> 6) << 4);
      mn[j] = ((unsigned)q[j + 4] >> 4) | ((unsigned)(q[j] >> 6) << 4);
    }
  }
  for (int p = 0; p < 4; p++) {
    const unsigned f = sc[2 * p] | (sc[2 * p + 1] << 6) | (mn[2 * p] << 12) | (mn[2 * p + 1] << 18);
    q[3 * p] = (unsigned char)(f & 0xffu);
    q[3 * p + 1] = (unsigned char)((f >> 8) & 0xffu);
    q[3 * p + 2] = (unsigned char)(f >> 16);
  }
}
// Q4_K: the qs plane class-major, in place (common.hpp): one thread per 32-byte chunk, byte 4 l + k <- byte 8 k + l.  A byte move:
// every nibble keeps its value; dequantized rows, block dots and the GEMM address the plane through q4k_perm_index.
__global__ __launch_bounds__(256) void k_q4k_class_major(unsigned* __restrict__ qs, size_t chunk0, size_t n_chunks) {
  const size_t c = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= n_chunks) return;
  unsigned* p = qs + (chunk0 + c) * 8;
  unsigned in[8], out[8];
#pragma unroll
  for (int i = 0; i < 8; i++) in[i] = p[i];
#pragma unroll
  for (int l = 0; l < 8; l++) {
    unsigned v = 0;
#pragma unroll
    for (int k = 0; k < 4; k++) {
      const int e = 8 * k + l;  // source byte
      v |= ((in[e >> 2] >> (8 * (e & 3))) & 0xffu) << (8 * k);
    }
    out[l] = v;
  }
#pragma unroll
  for (int i = 0; i < 8; i++) p[i] = out[i];
}
void launch_q4k_class_major(hipStream_t st, void* qs_plane, size_t blk0, size_t n_blocks) {
  if (n_blocks == 0) return;
  const size_t n_chunks = n_blocks * 4;
  k_q4k_class_major<<<(unsigned)((n_chunks + 255) / 256), 256, 0, st>>>((unsigned*)qs_plane, blk0 * 4, n_chunks);
}
void launch_q4k_pack_scales(hipStream_t st, void* hdr_plane, size_t blk0, size_t n_blocks) {
  if (n_blocks == 0) return;
  k_q4k_pack_scales<<<(unsigned)((n_blocks + 255) / 256), 256, 0, st>>>((unsigned char*)hdr_plane, blk0, n_blocks);
}

// ---- read ceiling: what a plain streaming kernel reads per second on this box (the practical HBM ceiling) ----------
// 16-byte non-temporal loads; the xor keeps the loads alive.  Three access patterns, the caller keeps the best (round-2 verdict:
// the grid-stride pattern alone read 6.2 TB/s on a box where the product's own classifier GEMV streamed 7.05 TB/s -- not a ceiling):
//   0  grid-stride, 8 loads in flight per lane (each lane's loads are gridDim x 4 KiB apart)
//   1  the GEMV's pattern: short-lived 128-thread workgroups, every wave reads ONE contiguous 4 KiB span (4 x 1 KiB requests,
//      all in flight) and exits -- what k_gemv_q4_0<2> does with two 2304-byte rows per wave
//   2  persistent: 2048 workgroups x 256 threads, every wave walks contiguous 8 KiB spans, 8 requests in flight
__global__ __launch_bounds__(256) void k_stream_read(const i32x4* __restrict__ p, size_t n16, int* __restrict__ sink) {
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  int acc = 0;
  for (; i + 7 * stride < n16; i += 8 * stride) {
    i32x4 v[8];
#pragma unroll
    for (int u = 0; u < 8; u++) v[u] = __builtin_nontemporal_load(p + i + u * stride);
#pragma unroll
    for (int u = 0; u < 8; u++) acc ^= v[u][0] ^ v[u][1] ^ v[u][2] ^ v[u][3];
  }
  for (; i < n16; i += stride) {
    const i32x4 v = __builtin_nontemporal_load(p + i);
    acc ^= v[0] ^ v[3];
  }
  if (acc == 0x7eadbeef) *sink = acc;
}
__global__ __launch_bounds__(128) void k_stream_read_spans(const i32x4* __restrict__ p, size_t n16, int* __restrict__ sink) {
  const size_t wave = (size_t)blockIdx.x * 2 + (threadIdx.x >> 6);
  const size_t i = wave * 256 + (threadIdx.x & 63);  // 256 16-byte units = 4 KiB per wave
  if (i + 192 >= n16) return;
  i32x4 v[4];
#pragma unroll
  for (int u = 0; u < 4; u++) v[u] = __builtin_nontemporal_load(p + i + u * 64);
  int acc = 0;
#pragma unroll
  for (int u = 0; u < 4; u++) acc ^= v[u][0] ^ v[u][1] ^ v[u][2] ^ v[u][3];
  if (acc == 0x7eadbeef) *sink = acc;
}
__global__ __launch_bounds__(256) void k_stream_read_persistent(const i32x4* __restrict__ p, size_t n16, int* __restrict__ sink) {
  const size_t nwaves = (size_t)gridDim.x * 4, wave = (size_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  int acc = 0;
  for (size_t s = wave * 512; s + 512 <= n16; s += nwaves * 512) {  // 512 units = 8 KiB per wave and step
    i32x4 v[8];
#pragma unroll
    for (int u = 0; u < 8; u++) v[u] = __builtin_nontemporal_load(p + s + u * 64 + (threadIdx.x & 63));
#pragma unroll
    for (int u = 0; u < 8; u++) acc ^= v[u][0] ^ v[u][1] ^ v[u][2] ^ v[u][3];
  }
  if (acc == 0x7eadbeef) *sink = acc;
}
void launch_stream_read(hipStream_t st, const void* buf, size_t bytes, int* sink, hipEvent_t e0, hipEvent_t e1, int pattern) {
  const size_t n16 = bytes / 16;
  if (pattern == 1)
    hipExtLaunchKernelGGL(k_stream_read_spans, dim3((unsigned)(n16 / 512)), dim3(128), 0, st, e0, e1, 0, (const i32x4*)buf, n16, sink);
  else if (pattern == 2)
    hipExtLaunchKernelGGL(k_stream_read_persistent, dim3(2048), dim3(256), 0, st, e0, e1, 0, (const i32x4*)buf, n16, sink);
  else
    hipExtLaunchKernelGGL(k_stream_read, dim3(8192), dim3(256), 0, st, e0, e1, 0, (const i32x4*)buf, n16, sink);
}

}  // namespace crabml_hip

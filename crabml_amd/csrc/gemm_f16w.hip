// gemm_f16w.hip -- the FAST prompt pass's weight GEMM (crabml_hip_llama_prefill on the fast device; Q4_0 weights, Q8_0 rows):
// weight-stationary on the f16 matrix cores.
//
// matmul_vec with a batched rhs is `C[b, m] = W[m, k] . x[b, k]` (matmul_vec.rs:41-76: the (b, k) rhs contract; llama2.rs:111-129).
// The bit-exact form (gemm_mfma.hip: exact int8 tiles, then the reference's per-block `sumf += (sumi as f32 * d_w) * d_x`) pays
// 8-10 VALU operations per MFMA for that scaling -- it is VALU-issue-bound at 14 % MfmaUtil (profiles/r05_prefill_gemm_experiments.md).
// Here the block scales are folded into the OPERANDS and the sum runs in f32 inside the matrix core across all blocks:
//   A' = (q_w - 8) * d_w   as f16: q_w - 8 exact, ONE rounding of the product (v_pk_mul_f16)
//   B' = q_x * d_x         as f16: one rounding (k_q8_0_rows_to_f16: once per activation matrix, not per weight row tile)
//   C  = sum_k A' B'       in f32 (v_mfma_f32_16x16x32_f16), no per-block work at all.
// A deviation of the fast tier only (two extra f16 roundings per product against the reference's exact integer block dots; inside
// FAST_TOL, tests/test_hip_prefill.py); matmul_vec and the strict-order device keep the bit-exact int8 kernel.
//
// Tiling (what round 5's f16 experiment lacked): a wave owns F 16-row fragments x T = 8 column tiles of 16 prompt rows and unpacks
// each A fragment ONCE per 32 k-slots -- 15 VALU operations -- for T MFMAs: 15 F / (F T) < 2 VALU per MFMA, 0.9 with the staging
// amortized (the int8 kernel: 10.75).  A fragments come straight from global memory in MFMA layout: lane (i, g) loads the whole
// 16-byte block kb0 + g of row i (64 contiguous bytes per row and chunk) and step s = 0..3 of the chunk feeds dword s of every
// lane -- an MFMA's 32 k-slots then span FOUR blocks (8 elements of each), which the folded scales allow.  B' tiles (128 columns x
// 128 k-slots x 2 B = 32 KB per chunk) go through LDS, double-buffered, one barrier per chunk, shared by the workgroup's four waves.
#include <type_traits>

#include "devutil.hpp"
#include "kernels.hpp"

namespace crabml_hip {

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));

// ---- B': the Q8_0 rows of a prompt pass as pre-scaled f16, in the GEMM's k-slot order ---------------------------------------
// xh[col][kb][32] f16; inside a block, slot 8 s + e (s = 0..3 = the dword of the weight block the slot pairs with) holds element
//   e = 0, 1: 4 s, 4 s + 2     e = 2, 3: 4 s + 1, 4 s + 3     e = 4, 5: 16 + 4 s, 16 + 4 s + 2     e = 6, 7: 16 + 4 s + 1, 16 + 4 s + 3
// -- the order in which unpack_q4_0_f16 below takes the nibbles out of a dword (two masks per packed pair, no byte permute).
__device__ __forceinline__ int f16w_slot_elem(int slot) {
  const int s = slot >> 3, e = slot & 7;
  return (e >= 4 ? 16 : 0) + 4 * s + ((e >> 1) & 1) + 2 * (e & 1);
}
__global__ __launch_bounds__(256) void k_q8_0_rows_to_f16(const char* __restrict__ planes, size_t row_stride, size_t off_d, int nb,
                                                          unsigned short* __restrict__ xh) {
  const size_t col = blockIdx.y;
  const int t = blockIdx.x * blockDim.x + threadIdx.x;  // one thread per (block, 8-slot group)
  if (t >= nb * 4) return;
  const int kb = t >> 2, s = t & 3;
  const char* p = planes + col * row_stride;
  const float d = h2f(((const unsigned short*)(p + off_d))[kb]);
  const signed char* q = (const signed char*)p + kb * 32;
  unsigned short o[8];
#pragma unroll
  for (int e = 0; e < 8; e++) o[e] = f2h((float)q[f16w_slot_elem(8 * s + e)] * d);  // (7-bit x 11-bit: exact in f32, one rounding)
  unsigned short* dst = xh + (col * nb + kb) * 32 + 8 * s;
  *(i32x4*)dst = i32x4{(int)(o[0] | ((unsigned)o[1] << 16)), (int)(o[2] | ((unsigned)o[3] << 16)), (int)(o[4] | ((unsigned)o[5] << 16)),
                       (int)(o[6] | ((unsigned)o[7] << 16))};
}
void launch_q8_0_rows_to_f16(hipStream_t st, const void* planes, size_t row_stride, size_t off_d, size_t rows, size_t k, void* xh) {
  const int nb = (int)(k / 32);
  k_q8_0_rows_to_f16<<<dim3((unsigned)((nb * 4 + 255) / 256), (unsigned)rows), 256, 0, st>>>((const char*)planes, row_stride, off_d, nb,
                                                                                               (unsigned short*)xh);
}

// one dword of a Q4_0 block (quant bytes 4 s .. 4 s + 3: low nibbles = elements 4 s .., high nibbles = 16 + 4 s ..; buf_q4_0.rs:24-33)
// -> the lane's eight f16 k-slots (q - 8) * d.  0x6400 | n is the f16 number 1024 + n; -1032 makes it n - 8 exactly.
__device__ __forceinline__ f16x8 unpack_q4_0_f16(unsigned w, f16x2 d2) {
  const f16x2 bias = {(_Float16)-1032.0f, (_Float16)-1032.0f};
  const unsigned u0 = (w & 0x000F000Fu) | 0x64006400u, u1 = ((w >> 8) & 0x000F000Fu) | 0x64006400u;
  const unsigned u2 = ((w >> 4) & 0x000F000Fu) | 0x64006400u, u3 = ((w >> 12) & 0x000F000Fu) | 0x64006400u;
  const f16x2 p0 = (__builtin_bit_cast(f16x2, u0) + bias) * d2, p1 = (__builtin_bit_cast(f16x2, u1) + bias) * d2;
  const f16x2 p2 = (__builtin_bit_cast(f16x2, u2) + bias) * d2, p3 = (__builtin_bit_cast(f16x2, u3) + bias) * d2;
  return f16x8{p0[0], p0[1], p1[0], p1[1], p2[0], p2[1], p3[0], p3[1]};
}

struct GemmF16Geo {
  static constexpr int T = 8, CW = 16 * T;     // column tiles per wave / prompt rows per workgroup
  static constexpr int KCH = 4;                // blocks per chunk
  static constexpr int CSTR = KCH * 64 + 16;   // bytes per column in an LDS buffer: 256 + 16 of padding (fragment reads: 16 lanes of a
                                               // tile, four banks each, cover the 64 banks once)
  static constexpr int BUF = CW * CSTR, LDS_BYTES = 2 * BUF;
  static constexpr int B_LOADS = CW * KCH * 4 / 256;  // 16-byte pieces per thread and chunk (8)
};

template <int F>  // 16-row fragments per wave: the workgroup's four waves own 64 F consecutive weight rows
__global__ __launch_bounds__(256, 2) void k_gemm_f16w(const i32x4* __restrict__ wq, const unsigned short* __restrict__ wd,
                                                      const i32x4* __restrict__ xh, float* __restrict__ out, int m, int nb, int n,
                                                      int row_tiles) {
  using G = GemmF16Geo;
  constexpr int T = G::T, KCH = G::KCH;
  extern __shared__ __attribute__((aligned(16))) unsigned char f16w_lds[];
  const int tid = threadIdx.x, lane = tid & 63, wave = wave_in_wg();
  const int i = lane & 15, g = lane >> 4;
  int rt, ct;
  {  // XCD-aware tile order (gemm_mfma.hip): the column tiles of a weight row tile back to back on ONE XCD
    const int col_tiles = (int)gridDim.x / row_tiles;
    if ((row_tiles & 7) == 0) {
      const int x = (int)blockIdx.x & 7, j = (int)blockIdx.x >> 3;
      ct = j % col_tiles;
      rt = (j / col_tiles) * 8 + x;
    } else {
      rt = (int)blockIdx.x % row_tiles;
      ct = (int)blockIdx.x / row_tiles;
    }
  }
  const int r0 = rt * 64 * F + wave * 16 * F, c0 = ct * G::CW;
  const int nchunks = (nb + KCH - 1) / KCH;

  // A: the lane's block (row i of fragment f, block kb0 + g) and its scale.  HBM latency is several chunk times (a chunk is ~0.4 us of
  // MFMAs and a workgroup has the SIMD almost to itself): a RING of four register sets, chunk c + 3 requested while chunk c is
  // multiplied; B' (L2-resident) two chunks ahead in two register sets.  All ring indices are compile-time (chunk loop unrolled by 4).
  i32x4 aq[4][F];
  unsigned ad[4][F];
  auto fetch_a = [&](i32x4 (&q)[F], unsigned (&d)[F], int ch) {
    const int cc = ch < nchunks ? ch : nchunks - 1;  // (past the end: re-read the last chunk, never consumed)
    const int kb = cc * KCH + g;
    const int gkb = kb < nb ? kb : nb - 1;
#pragma unroll
    for (int f = 0; f < F; f++) {
      const int row = r0 + 16 * f + i;
      const size_t blk = (size_t)(row < m ? row : m - 1) * nb + gkb;
      q[f] = __builtin_nontemporal_load(wq + blk);
      const unsigned dv = __builtin_nontemporal_load(wd + blk);
      d[f] = kb < nb ? dv : 0u;  // past the row's end: scale 0, the slots add nothing
    }
  };
  // B': 128 columns x 256 bytes per chunk, 8 pieces per thread (piece p: column p / 16, 16 bytes p % 16 of the chunk)
  i32x4 rb[2][G::B_LOADS];
  auto fetch_b = [&](i32x4 (&r)[G::B_LOADS], int ch) {
    const int cc = ch < nchunks ? ch : nchunks - 1;
#pragma unroll
    for (int u = 0; u < G::B_LOADS; u++) {
      const int p = tid + 256 * u, col = p >> 4, pc = p & 15;
      const int gcol = c0 + col < n ? c0 + col : n - 1;
      const int kb = cc * KCH + (pc >> 2);
      const int gkb = kb < nb ? kb : nb - 1;  // (the tail chunk re-reads the last block: finite values against zero weights)
      r[u] = xh[((size_t)gcol * nb + gkb) * 4 + (pc & 3)];
    }
  };
  auto commit_b = [&](const i32x4 (&r)[G::B_LOADS], int buf) {
    unsigned char* S = f16w_lds + buf * G::BUF;
#pragma unroll
    for (int u = 0; u < G::B_LOADS; u++) {
      const int p = tid + 256 * u, col = p >> 4, pc = p & 15;
      *(i32x4*)(S + col * G::CSTR + pc * 16) = r[u];
    }
  };

  f32x4 acc[F][T];
#pragma unroll
  for (int f = 0; f < F; f++)
#pragma unroll
    for (int t = 0; t < T; t++) acc[f][t] = f32x4{0.f, 0.f, 0.f, 0.f};

  fetch_b(rb[0], 0);
  fetch_a(aq[0], ad[0], 0);
  fetch_a(aq[1], ad[1], 1);
  fetch_a(aq[2], ad[2], 2);
  commit_b(rb[0], 0);
  fetch_b(rb[1], 1);  // chunk c + 1 sits in rb[(c + 1) & 1] when chunk c is multiplied
  fetch_b(rb[0], 2);
  __syncthreads();
  // chunk ch (ring slot J, LDS buffer ch & 1)
  auto chunk = [&](auto Jc, int ch) {
    constexpr int J = decltype(Jc)::value;
    fetch_a(aq[(J + 3) & 3], ad[(J + 3) & 3], ch + 3);
    const unsigned char* S = f16w_lds + (J & 1) * G::BUF + i * G::CSTR + g * 64;
    f16x2 d2[F];
#pragma unroll
    for (int f = 0; f < F; f++) d2[f] = __builtin_bit_cast(f16x2, ad[J][f] | (ad[J][f] << 16));
#pragma unroll
    for (int s = 0; s < 4; s++) {
      f16x8 a[F];
#pragma unroll
      for (int f = 0; f < F; f++) a[f] = unpack_q4_0_f16((unsigned)aq[J][f][s], d2[f]);
#pragma unroll
      for (int t = 0; t < T; t++) {
        const f16x8 b = *(const f16x8*)(S + t * 16 * G::CSTR + s * 16);
#pragma unroll
        for (int f = 0; f < F; f++) acc[f][t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[f], b, acc[f][t], 0, 0, 0);
      }
    }
    commit_b(rb[(J + 1) & 1], (J + 1) & 1);  // chunk ch + 1 into the other buffer (read last in iteration ch - 1: behind its barrier)
    fetch_b(rb[(J + 1) & 1], ch + 3);        // ... and the freed register set takes chunk ch + 3
    __syncthreads();
  };
  for (int ch = 0; ch < nchunks; ch += 4) {  // (uniform conditions: every thread takes the barrier inside a chunk or none does)
    chunk(std::integral_constant<int, 0>{}, ch);
    if (ch + 1 < nchunks) chunk(std::integral_constant<int, 1>{}, ch + 1);
    if (ch + 2 < nchunks) chunk(std::integral_constant<int, 2>{}, ch + 2);
    if (ch + 3 < nchunks) chunk(std::integral_constant<int, 3>{}, ch + 3);
  }
  // D: lane (i, g) holds rows 4 g .. 4 g + 3 of column i of every tile
#pragma unroll
  for (int t = 0; t < T; t++) {
    const int col = c0 + 16 * t + i;
    if (col >= n) continue;
#pragma unroll
    for (int f = 0; f < F; f++) {
      const int row = r0 + 16 * f + 4 * g;
      float* o = out + (size_t)col * m + row;
      if (row + 3 < m) {
        *(f32x4*)o = acc[f][t];
      } else {
#pragma unroll
        for (int r = 0; r < 4; r++)
          if (row + r < m) o[r] = acc[f][t][r];
      }
    }
  }
}

// xh: the rows' pre-scaled f16 planes (launch_q8_0_rows_to_f16); returns false when the shape is not covered
bool launch_gemm_f16w(crabml_hip_device* dev, const crabml_hip_buf* w, size_t m, size_t k, const void* xh, size_t b, float* out) {
  if (w->dtype != CRABML_HIP_Q4_0 || k % 32 != 0 || m % 4 != 0 || b < 32) return false;
  using G = GemmF16Geo;
  hipStream_t st = dev->stream;
  const char* wp = (const char*)w->ptr;
  const int nb = (int)(k / 32);
  const int col_tiles = (int)((b + G::CW - 1) / G::CW);
  // 128-row workgroups (two fragments per wave: every B' fragment read from LDS feeds two MFMAs) when that still covers the chip
  static bool raised = false;
  if (!raised) {
    if (hipFuncSetAttribute((const void*)k_gemm_f16w<1>, hipFuncAttributeMaxDynamicSharedMemorySize, G::LDS_BYTES) != hipSuccess ||
        hipFuncSetAttribute((const void*)k_gemm_f16w<2>, hipFuncAttributeMaxDynamicSharedMemorySize, G::LDS_BYTES) != hipSuccess) {
      (void)hipGetLastError();
      return false;
    }
    raised = true;
  }
  const bool wide = ((m + 127) / 128) * (size_t)col_tiles >= (size_t)dev->n_cu;
  if (wide) {
    const int row_tiles = (int)((m + 127) / 128);
    k_gemm_f16w<2><<<dim3(row_tiles * col_tiles), 256, G::LDS_BYTES, st>>>((const i32x4*)wp, (const unsigned short*)(wp + w->wl.off_scale),
                                                                          (const i32x4*)xh, out, (int)m, nb, (int)b, row_tiles);
  } else {
    const int row_tiles = (int)((m + 63) / 64);
    k_gemm_f16w<1><<<dim3(row_tiles * col_tiles), 256, G::LDS_BYTES, st>>>((const i32x4*)wp, (const unsigned short*)(wp + w->wl.off_scale),
                                                                          (const i32x4*)xh, out, (int)m, nb, (int)b, row_tiles);
  }
  return true;
}

}  // namespace crabml_hip

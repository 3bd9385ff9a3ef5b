// gemm_mfma.hip -- matmul_vec with a batched rhs (b >= 16 rows of activations): skinny GEMM on the matrix cores.
//
// The reference runs `C[b, m] = W[m, k] . x[b, k]` as b independent dots (matmul_vec.rs:41-76); every weight row is
// streamed once per batch row.  Here one wavefront owns 16 weight rows and up to 64 batch rows and streams the
// weights ONCE: per 32-element block one `v_mfma_i32_16x16x32_i8` per 16 batch rows gives the exact integer dots
// sum_k w[i][k] x[j][k] of the 16 x 16 tile, which are then scaled exactly as the reference's scalar loop does,
// `sumf += (sumi as f32 * d_w) * d_x` (buf_q4_0.rs:249, buf_q8_0.rs:282), block after block in order -- so each
// output equals the reference's scalar-order result bit for bit (the per-block scales forbid accumulating across
// blocks inside the MFMA, which also bounds its utilisation: 4 conversions + 8 f32 ops per lane per MFMA).
//
// MFMA operand layout (16x16x32 i8): lane l supplies, for A, 8 k-slots of row i = l & 15 and, for B, the same 8
// k-slots of column j = l & 15; k-slot group = l >> 4.  The product is invariant under any permutation of the 32
// k-slots applied to both operands, so the slots are assigned for load convenience:
//   Q4_0: lane group g takes the block's quant bytes [4g, 4g+4): low nibbles = elements 4g..4g+3, high nibbles =
//         elements 16+4g..16+4g+3 (buf_q4_0.rs:24-33); the -8 offset is applied as -8 * sum(x) per block (exact);
//   Q8_0: lane group g takes elements [8g, 8g+8).
// D: lane l holds rows (l >> 4) * 4 + r (r = 0..3) of column l & 15.
#include "devutil.hpp"
#include "gemv_core.hpp"
#include "kernels.hpp"

namespace crabml_hip {

// Workgroup = 4 waves = 64 weight rows x 64 batch columns; k runs in chunks of KC = 8 blocks staged through LDS:
// the weights with coalesced 16-byte loads (one pass over HBM), the activation planes from L2 (shared by the four
// waves).  LDS words are laid out [block][dword of the block][row or column, padded to 72]: the fragment reads of
// a wave (lane = (i, g): dword g / 4 + g of row / column i) then hit every bank exactly twice.
template <int FMT>
struct GemmGeo {
  static constexpr int KC = 8;                                  // blocks per chunk
  static constexpr int WPB = FMT == CRABML_HIP_Q4_0 ? 4 : 8;    // dwords of quants per weight block
  static constexpr int PAD = 72;
  static constexpr int A_WORDS = KC * WPB * PAD;
  static constexpr int B_WORDS = KC * 8 * PAD;
  // bytes: A quants | B quants | A scales f16 [KC][64] | B scales f16 [KC][64] | B isum i32 [KC][64]
  static constexpr int LDS_BYTES = (A_WORDS + B_WORDS) * 4 + KC * 64 * 2 * 2 + KC * 64 * 4;
};

template <int FMT>
__global__ __launch_bounds__(256) void k_gemm_mfma(const char* __restrict__ wq, const unsigned short* __restrict__ wd,
                                                   const char* __restrict__ act, size_t act_stride, size_t off_d, size_t off_aux,
                                                   float* __restrict__ out, int m, int nb, int b, int row_tiles) {
  using G = GemmGeo<FMT>;
  constexpr int KC = G::KC, WPB = G::WPB, PAD = G::PAD;
  extern __shared__ unsigned lds_w[];
  unsigned* sA = lds_w;
  unsigned* sB = sA + G::A_WORDS;
  unsigned short* sAd = (unsigned short*)(sB + G::B_WORDS);
  unsigned short* sBd = sAd + KC * 64;
  int* sBs = (int*)(sBd + KC * 64);
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int r0 = (blockIdx.x % row_tiles) * 64, c0 = (blockIdx.x / row_tiles) * 64;
  const int i = lane & 15, g = lane >> 4;

  float F[4][4];
#pragma unroll
  for (int jt = 0; jt < 4; jt++)
#pragma unroll
    for (int r = 0; r < 4; r++) F[jt][r] = 0.0f;

  for (int kb0 = 0; kb0 < nb; kb0 += KC) {
    const int kc = nb - kb0 < KC ? nb - kb0 : KC;
    __syncthreads();  // the previous chunk has been consumed
    // ---- stage A: 64 rows x kc blocks; a thread moves 16 bytes (one Q4_0 block, half a Q8_0 block)
    constexpr int PIECES = WPB / 4;  // 16-byte pieces per block
    for (int t = tid; t < 64 * KC * PIECES; t += 256) {
      const int row = t / (KC * PIECES), rem = t % (KC * PIECES), kb = rem / PIECES, pc = rem % PIECES;
      if (kb < kc) {
        const int grow = r0 + row < m ? r0 + row : m - 1;
        const i32x4 v = __builtin_nontemporal_load((const i32x4*)wq + ((size_t)grow * nb + kb0 + kb) * PIECES + pc);
#pragma unroll
        for (int q = 0; q < 4; q++) sA[(kb * WPB + pc * 4 + q) * PAD + row] = (unsigned)v[q];
      }
    }
    for (int t = tid; t < 64 * KC; t += 256) {
      const int row = t / KC, kb = t % KC;
      if (kb < kc) {
        const int grow = r0 + row < m ? r0 + row : m - 1;
        sAd[kb * 64 + row] = wd[(size_t)grow * nb + kb0 + kb];
      }
    }
    // ---- stage B: 64 columns x kc blocks x 32 int8 (two 16-byte pieces), scales, block sums
    for (int t = tid; t < 64 * KC * 2; t += 256) {
      const int col = t / (KC * 2), rem = t % (KC * 2), kb = rem / 2, pc = rem % 2;
      if (kb < kc) {
        const int gcol = c0 + col < b ? c0 + col : b - 1;
        const i32x4 v = *((const i32x4*)(act + (size_t)gcol * act_stride) + (size_t)(kb0 + kb) * 2 + pc);
#pragma unroll
        for (int q = 0; q < 4; q++) sB[(kb * 8 + pc * 4 + q) * PAD + col] = (unsigned)v[q];
      }
    }
    for (int t = tid; t < 64 * KC; t += 256) {
      const int col = t / KC, kb = t % KC;
      if (kb < kc) {
        const int gcol = c0 + col < b ? c0 + col : b - 1;
        const char* ap = act + (size_t)gcol * act_stride;
        sBd[kb * 64 + col] = ((const unsigned short*)(ap + off_d))[kb0 + kb];
        sBs[kb * 64 + col] = ((const int*)(ap + off_aux))[kb0 + kb];
      }
    }
    __syncthreads();
    // ---- compute: wave = rows 16 wave .. +16, all 4 column tiles
    for (int kb = 0; kb < kc; kb++) {
      long A;
      if (FMT == CRABML_HIP_Q4_0) {
        const unsigned w = sA[(kb * 4 + g) * PAD + 16 * wave + i];
        A = (long)(((unsigned long long)((w >> 4) & 0x0F0F0F0Fu) << 32) | (unsigned long long)(w & 0x0F0F0F0Fu));
      } else {
        const unsigned lo = sA[(kb * 8 + 2 * g) * PAD + 16 * wave + i], hi = sA[(kb * 8 + 2 * g + 1) * PAD + 16 * wave + i];
        A = (long)(((unsigned long long)hi << 32) | (unsigned long long)lo);
      }
      float dw[4];
#pragma unroll
      for (int r = 0; r < 4; r++) dw[r] = h2f(sAd[kb * 64 + 16 * wave + 4 * g + r]);
#pragma unroll
      for (int jt = 0; jt < 4; jt++) {
        const int col = 16 * jt + i;
        unsigned lo, hi;
        if (FMT == CRABML_HIP_Q4_0) {
          lo = sB[(kb * 8 + g) * PAD + col];
          hi = sB[(kb * 8 + 4 + g) * PAD + col];
        } else {
          lo = sB[(kb * 8 + 2 * g) * PAD + col];
          hi = sB[(kb * 8 + 2 * g + 1) * PAD + col];
        }
        const long B = (long)(((unsigned long long)hi << 32) | (unsigned long long)lo);
        // Q4_0: the -8 offset rides in as the accumulator input, -8 * sum(x) of the lane's column (exact)
        const int cin = FMT == CRABML_HIP_Q4_0 ? -8 * sBs[kb * 64 + col] : 0;
        const i32x4 D = __builtin_amdgcn_mfma_i32_16x16x32_i8(A, B, i32x4{cin, cin, cin, cin}, 0, 0, 0);
        const float dx = h2f(sBd[kb * 64 + col]);
#pragma unroll
        for (int r = 0; r < 4; r++) F[jt][r] += ((float)D[r] * dw[r]) * dx;
      }
    }
  }
#pragma unroll
  for (int jt = 0; jt < 4; jt++) {
    const int col = c0 + 16 * jt + i;
    if (col >= b) continue;
#pragma unroll
    for (int r = 0; r < 4; r++) {
      const int row = r0 + 16 * wave + g * 4 + r;
      if (row < m) out[(size_t)col * m + row] = F[jt][r];
    }
  }
}

// returns false when the shape / format is not covered (the caller falls back to one GEMV per batch row)
bool launch_gemm_mfma(crabml_hip_device* dev, const crabml_hip_buf* w, size_t m, size_t k, const void* act, size_t b, float* out,
                      crabml_hip_device::ProfRec* rec) {
  if (w->dtype != CRABML_HIP_Q4_0 && w->dtype != CRABML_HIP_Q8_0) return false;
  if (b < 16 || m == 0 || k % 32 != 0) return false;
  hipStream_t st = dev->stream;
  const char* wp = (const char*)w->ptr;
  const ActLayout al = act_layout(CRABML_HIP_Q8_0, k);
  const int nb = (int)(k / 32);
  const int row_tiles = (int)((m + 63) / 64), col_tiles = (int)((b + 63) / 64);
  if (w->dtype == CRABML_HIP_Q4_0)
    launch_k(st, rec, k_gemm_mfma<CRABML_HIP_Q4_0>, dim3(row_tiles * col_tiles), dim3(256), GemmGeo<CRABML_HIP_Q4_0>::LDS_BYTES, wp,
             (const unsigned short*)(wp + w->wl.off_scale), (const char*)act, al.total, al.off_d, al.off_aux, out, (int)m, nb, (int)b,
             row_tiles);
  else
    launch_k(st, rec, k_gemm_mfma<CRABML_HIP_Q8_0>, dim3(row_tiles * col_tiles), dim3(256), GemmGeo<CRABML_HIP_Q8_0>::LDS_BYTES, wp,
             (const unsigned short*)(wp + w->wl.off_scale), (const char*)act, al.total, al.off_d, al.off_aux, out, (int)m, nb, (int)b,
             row_tiles);
  return true;
}

}  // namespace crabml_hip

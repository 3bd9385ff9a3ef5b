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

// Workgroup = 4 waves = 64 weight rows x 64 batch columns; k runs in chunks of KC = 8 blocks staged through two LDS
// buffers: while a chunk is multiplied, the next one is already in flight from HBM (weights, coalesced 16-byte
// loads: one pass) and L2 (activation planes, shared by the four waves) into registers, and lands in the other
// buffer behind a single barrier per chunk.  LDS rows are [block][row or column][RW words] with RW = 4 (Q4_0
// quants) or 12 (8 data words + 4 pad): staging writes are 16-byte vectors and a wave's fragment reads (lane =
// (i, g): words g / 4 + g, or 2g / 2g + 1, of row i) hit every bank exactly twice.
template <int FMT>
struct GemmGeo {
  static constexpr int KC = 4;                                     // blocks per chunk (37 KB of LDS per workgroup: 4 workgroups per CU)
  static constexpr int APC = FMT == CRABML_HIP_Q4_0 ? 1 : 2;       // 16-byte pieces per weight block
  static constexpr int ARW = FMT == CRABML_HIP_Q4_0 ? 4 : 12;      // LDS words per weight block row
  static constexpr int BRW = 12;                                   // LDS words per activation block row
  static constexpr int A_WORDS = KC * 64 * ARW, B_WORDS = KC * 64 * BRW;
  // one buffer: A quants | B quants | A scales f16 [KC][64] | B scales f16 [KC][64] | B isum i32 [KC][64]
  static constexpr int BUF_BYTES = (A_WORDS + B_WORDS) * 4 + KC * 64 * 2 * 2 + KC * 64 * 4;
  static constexpr int LDS_BYTES = 2 * BUF_BYTES;
  static constexpr int A_LOADS = 64 * KC * APC / 256, B_LOADS = 64 * KC * 2 / 256, S_LOADS = 64 * KC / 256;
};

template <int FMT>
__global__ __launch_bounds__(256) void k_gemm_mfma(const char* __restrict__ wq, const unsigned short* __restrict__ wd,
                                                   const char* __restrict__ act, size_t act_stride, size_t off_d, size_t off_aux,
                                                   float* __restrict__ out, int m, int nb, int b, int row_tiles) {
  using G = GemmGeo<FMT>;
  constexpr int KC = G::KC, APC = G::APC, ARW = G::ARW, BRW = G::BRW;
  extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int r0 = (blockIdx.x % row_tiles) * 64, c0 = (blockIdx.x / row_tiles) * 64;
  const int i = lane & 15, g = lane >> 4;

  // staging registers of one chunk
  i32x4 ra[G::A_LOADS], rb[G::B_LOADS];
  unsigned short rad[G::S_LOADS], rbd[G::S_LOADS];
  int rbs[G::S_LOADS];
  auto fetch = [&](int kb0) {
#pragma unroll
    for (int u = 0; u < G::A_LOADS; u++) {
      const int t = tid + 256 * u, row = t / (KC * APC), rem = t % (KC * APC), kb = rem / APC, pc = rem % APC;
      const int grow = r0 + row < m ? r0 + row : m - 1;
      const int gkb = kb0 + kb < nb ? kb0 + kb : nb - 1;  // the tail chunk re-reads the last block (never consumed)
      ra[u] = __builtin_nontemporal_load((const i32x4*)wq + ((size_t)grow * nb + gkb) * APC + pc);
    }
#pragma unroll
    for (int u = 0; u < G::B_LOADS; u++) {
      const int t = tid + 256 * u, col = t / (KC * 2), rem = t % (KC * 2), kb = rem / 2, pc = rem % 2;
      const int gcol = c0 + col < b ? c0 + col : b - 1;
      const int gkb = kb0 + kb < nb ? kb0 + kb : nb - 1;
      rb[u] = *((const i32x4*)(act + (size_t)gcol * act_stride) + (size_t)gkb * 2 + pc);
    }
#pragma unroll
    for (int u = 0; u < G::S_LOADS; u++) {
      const int t = tid + 256 * u, rc = t / KC, kb = t % KC;
      const int grow = r0 + rc < m ? r0 + rc : m - 1, gcol = c0 + rc < b ? c0 + rc : b - 1;
      const int gkb = kb0 + kb < nb ? kb0 + kb : nb - 1;
      const char* ap = act + (size_t)gcol * act_stride;
      rad[u] = wd[(size_t)grow * nb + gkb];
      rbd[u] = ((const unsigned short*)(ap + off_d))[gkb];
      rbs[u] = ((const int*)(ap + off_aux))[gkb];
    }
  };
  auto commit = [&](int buf) {
    unsigned* sA = (unsigned*)(lds_raw + (size_t)buf * G::BUF_BYTES);
    unsigned* sB = sA + G::A_WORDS;
    unsigned short* sAd = (unsigned short*)(sB + G::B_WORDS);
    unsigned short* sBd = sAd + KC * 64;
    int* sBs = (int*)(sBd + KC * 64);
#pragma unroll
    for (int u = 0; u < G::A_LOADS; u++) {
      const int t = tid + 256 * u, row = t / (KC * APC), rem = t % (KC * APC), kb = rem / APC, pc = rem % APC;
      *(i32x4*)(sA + (kb * 64 + row) * ARW + pc * 4) = ra[u];
    }
#pragma unroll
    for (int u = 0; u < G::B_LOADS; u++) {
      const int t = tid + 256 * u, col = t / (KC * 2), rem = t % (KC * 2), kb = rem / 2, pc = rem % 2;
      *(i32x4*)(sB + (kb * 64 + col) * BRW + pc * 4) = rb[u];
    }
#pragma unroll
    for (int u = 0; u < G::S_LOADS; u++) {
      const int t = tid + 256 * u, rc = t / KC, kb = t % KC;
      sAd[kb * 64 + rc] = rad[u];
      sBd[kb * 64 + rc] = rbd[u];
      sBs[kb * 64 + rc] = rbs[u];
    }
  };

  float F[4][4];
#pragma unroll
  for (int jt = 0; jt < 4; jt++)
#pragma unroll
    for (int r = 0; r < 4; r++) F[jt][r] = 0.0f;

  fetch(0);
  commit(0);
  __syncthreads();
  int buf = 0;
  for (int kb0 = 0; kb0 < nb; kb0 += KC, buf ^= 1) {
    const int kc = nb - kb0 < KC ? nb - kb0 : KC;
    const bool more = kb0 + KC < nb;
    if (more) fetch(kb0 + KC);  // in flight while this chunk is multiplied
    const unsigned* sA = (const unsigned*)(lds_raw + (size_t)buf * G::BUF_BYTES);
    const unsigned* sB = sA + G::A_WORDS;
    const unsigned short* sAd = (const unsigned short*)(sB + G::B_WORDS);
    const unsigned short* sBd = sAd + KC * 64;
    const int* sBs = (const int*)(sBd + KC * 64);
    for (int kb = 0; kb < kc; kb++) {
      long A;
      const unsigned* arow = sA + (kb * 64 + 16 * wave + i) * ARW;
      if (FMT == CRABML_HIP_Q4_0) {
        const unsigned w = arow[g];
        A = (long)(((unsigned long long)((w >> 4) & 0x0F0F0F0Fu) << 32) | (unsigned long long)(w & 0x0F0F0F0Fu));
      } else {
        A = *(const long*)(arow + 2 * g);
      }
      float dw[4];
      {
        const unsigned long long d4 = *(const unsigned long long*)(sAd + kb * 64 + 16 * wave + 4 * g);
#pragma unroll
        for (int r = 0; r < 4; r++) dw[r] = h2f((unsigned short)(d4 >> (16 * r)));
      }
      // all four column tiles: fragments first, then the four MFMAs back to back (independent accumulators),
      // then the scaling -- a single wave per SIMD has nothing else to hide the LDS and MFMA latencies behind
      long Bf[4];
      int cin[4];
      float dx[4];
#pragma unroll
      for (int jt = 0; jt < 4; jt++) {
        const int col = 16 * jt + i;
        const unsigned* brow = sB + (kb * 64 + col) * BRW;
        if (FMT == CRABML_HIP_Q4_0)
          Bf[jt] = (long)(((unsigned long long)brow[4 + g] << 32) | (unsigned long long)brow[g]);
        else
          Bf[jt] = *(const long*)(brow + 2 * g);
        // Q4_0: the -8 offset rides in as the accumulator input, -8 * sum(x) of the lane's column (exact)
        cin[jt] = FMT == CRABML_HIP_Q4_0 ? -8 * sBs[kb * 64 + col] : 0;
        dx[jt] = h2f(sBd[kb * 64 + col]);
      }
      i32x4 D[4];
#pragma unroll
      for (int jt = 0; jt < 4; jt++)
        D[jt] = __builtin_amdgcn_mfma_i32_16x16x32_i8(A, Bf[jt], i32x4{cin[jt], cin[jt], cin[jt], cin[jt]}, 0, 0, 0);
#pragma unroll
      for (int jt = 0; jt < 4; jt++)
#pragma unroll
        for (int r = 0; r < 4; r++) F[jt][r] += ((float)D[jt][r] * dw[r]) * dx[jt];
    }
    if (more) commit(buf ^ 1);  // the other buffer was last read one iteration ago (barrier below)
    __syncthreads();
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

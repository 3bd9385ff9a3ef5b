// gemm_f16w.hip -- the FAST prompt pass's weight GEMM (crabml_hip_llama_prefill on the fast device; Q4_0 / Q8_0 weights x Q8_0 rows,
// Q4_1 x Q8_1, Q4_K / Q6_K x Q8_K): weight-stationary on the f16 matrix cores.
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
// 128 k-slots x 2 B = 32 KB per chunk) go through LDS, double-buffered, one barrier per chunk, shared by the workgroup's four waves;
// the fragment reads are conflict-free ds_read_b128 (layout: GemmF16Geo), a group of tiles ahead of the MFMAs that consume them.
//
// Q8_0 weights: the lane's block is 32 bytes (two loads); step s takes dwords s and 4 + s -- the same k-slot order as Q4_0's nibbles,
// so both formats share one B'.  A' = q_w * d_w (q_w exact).
// Q4_K weights (Q8_K rows): a chunk is HALF a super-block -- lane (i, g) loads the 16-byte class-major piece h of pair g (dword s =
// class 4 h + s: elements 8 k + 4 h + s of sub-blocks 2 g (low nibbles) and 2 g + 1 (high); common.hpp) and, once per super-block, the
// 16-byte header.  A' = n * fl16(d * sc) - fl16(dmin * m) (buf_q4_k.rs:225-263's d * sc * q - dmin * m, per element): n exact, the
// product exact inside ONE v_pk_fma_f16, so three f16 roundings per weight element; B' = q_x * d_x from the Q8_K planes in the
// matching slot order (k_rows_to_f16<1>).
// Q4_1 weights (Q8_1 rows): Q4_0's loads and slot order; A' = n * d + m in one v_pk_fma_f16 (buf_q4_1.rs: d * q + m per element --
// the rows' `s` term of the integer form is not needed).
// Q6_K weights (Q8_K rows): a chunk is one 128-element half of a super-block -- lane (i, g) loads 16 bytes of ql (g < 2: ql[0..31],
// the low nibbles are quarter 0, the high ones quarter 2; g >= 2: ql[32..63], quarters 1 and 3; buf_q6_k.rs:21-48) and the 16 bytes
// of qh that carry the same l's two high bits, and once per super-block the 16 int8 scales and d.  A' = (q - 32) * fl16(d * sc),
// q - 32 exact.  Its own B' slot order (k_rows_to_f16<2>): a Q4_K_M layer converts the rows once per order it needs.
#include <cstdlib>
#include <type_traits>

#include "devutil.hpp"
#include "f16w_rows.hpp"
#include "kernels.hpp"

namespace crabml_hip {

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));

// ---- B': the rows of a prompt pass as pre-scaled f16, in the GEMM's k-slot order: xh[col][kb][32] f16 ------------------------
// (the slot orders and the conversion of one 8-slot group: f16w_rows.hpp, shared with the row kernels that write B' themselves)
// Q8_K rows (Q4_K weights): slot group kb = 4 cc + g of chunk cc = (super-block, half h), step s = class l = 4 h + s of pair g;
// slot e: k = 0, 2, 1, 3 of sub-block 2 g (e < 4) / 2 g + 1 (e >= 4), element 8 k + l -- unpack_q4_k_f16's order
// Q6_K weights: chunk cc = (super-block, half), g, step s: element 128 half + 32 (g >> 1) + 16 (g & 1) + 4 s + k (e < 4), + 64 (e >= 4)
template <int ORDER>  // 0: Q8_0 / Q8_1 rows (block order); 1: Q8_K rows for Q4_K weights; 2: Q8_K rows for Q6_K weights
__global__ __launch_bounds__(256) void k_rows_to_f16(const char* __restrict__ planes, size_t row_stride, size_t off_d, int nb,
                                                     unsigned short* __restrict__ xh) {
  const size_t col = blockIdx.y;
  const int t = blockIdx.x * blockDim.x + threadIdx.x;  // one thread per (block, 8-slot group)
  if (t >= nb * 4) return;
  rows_to_f16_piece<ORDER>(planes + col * row_stride, off_d, t, xh + col * (size_t)nb * 32);
}
// planes: `rows` sets of activation planes (act_layout(act_qtype, k)); the slot order is the one the weight format w_dtype reads
int gemm_f16w_order(uint32_t w_dtype) { return w_dtype == CRABML_HIP_Q4_K ? 1 : w_dtype == CRABML_HIP_Q6_K ? 2 : 0; }
bool launch_rows_to_f16(hipStream_t st, uint32_t act_qtype, uint32_t w_dtype, const void* planes, size_t rows, size_t k, void* xh) {
  if (!gemm_f16w_covers(w_dtype, act_qtype) || (act_qtype == CRABML_HIP_Q8_K && k % 256 != 0)) return false;
  const ActLayout al = act_layout(act_qtype, k);
  const int nb = (int)(k / 32);
  const dim3 grid((unsigned)((nb * 4 + 255) / 256), (unsigned)rows);
  switch (gemm_f16w_order(w_dtype)) {
    case 0: k_rows_to_f16<0><<<grid, 256, 0, st>>>((const char*)planes, al.total, al.off_d, nb, (unsigned short*)xh); break;
    case 1: k_rows_to_f16<1><<<grid, 256, 0, st>>>((const char*)planes, al.total, al.off_d, nb, (unsigned short*)xh); break;
    default: k_rows_to_f16<2><<<grid, 256, 0, st>>>((const char*)planes, al.total, al.off_d, nb, (unsigned short*)xh); break;
  }
  return true;
}

// the k pieces of a split GEMM: out[j] += part[j][0] + part[j][1] + ... in piece order (deterministic), every matrix of the launch at once
struct F16wParts {
  float* out[3];
  const float* part[3];  // matrix j's first partial; the next pieces follow pstride floats apart
  size_t end4[3];        // cumulative f32x4 counts
  size_t pstride;
  int nparts;
};
__global__ __launch_bounds__(256) void k_addn_f32(F16wParts a) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= a.end4[2]) return;
  const int j = i < a.end4[0] ? 0 : i < a.end4[1] ? 1 : 2;
  const size_t li = i - (j == 0 ? 0 : j == 1 ? a.end4[0] : a.end4[1]);
  float* o = j == 0 ? a.out[0] : j == 1 ? a.out[1] : a.out[2];
  const float* p = j == 0 ? a.part[0] : j == 1 ? a.part[1] : a.part[2];
  f32x4 v = ((const f32x4*)o)[li];
  for (int s = 0; s < a.nparts; s++) v = v + ((const f32x4*)(p + (size_t)s * a.pstride))[li];
  ((f32x4*)o)[li] = v;
}

// one dword of a Q4_0 block (quant bytes 4 s .. 4 s + 3: low nibbles = elements 4 s .., high nibbles = 16 + 4 s ..; buf_q4_0.rs:24-33)
// -> the lane's eight f16 k-slots (q - 8) * d.  0x6400 | n is the f16 number 1024 + n; -1032 makes it n - 8 exactly.
// (x & 0x000F000F) | 0x64006400 as ONE v_and_or_b32: a VOP3 instruction takes a single scalar / literal operand, so the compiler
// splits it unless one constant sits in a VGPR -- `magic` (loop-invariant, one register)
__device__ __forceinline__ unsigned and_or_magic(unsigned x, unsigned magic) {
  unsigned r;
  asm("v_and_or_b32 %0, %1, %2, %3" : "=v"(r) : "v"(x), "s"(0x000F000Fu), "v"(magic));
  return r;
}
__device__ __forceinline__ f16x8 unpack_q4_0_f16(unsigned w, f16x2 d2, unsigned magic) {
  const f16x2 bias = {(_Float16)-1032.0f, (_Float16)-1032.0f};
  const unsigned u0 = and_or_magic(w, magic), u1 = and_or_magic(w >> 8, magic);
  const unsigned u2 = and_or_magic(w >> 4, magic), u3 = and_or_magic(w >> 12, magic);
  const f16x2 p0 = (__builtin_bit_cast(f16x2, u0) + bias) * d2, p1 = (__builtin_bit_cast(f16x2, u1) + bias) * d2;
  const f16x2 p2 = (__builtin_bit_cast(f16x2, u2) + bias) * d2, p3 = (__builtin_bit_cast(f16x2, u3) + bias) * d2;
  return f16x8{p0[0], p0[1], p1[0], p1[1], p2[0], p2[1], p3[0], p3[1]};
}

// Q8_0: the bytes of a dword as f16 (b ^ 0x80 = b + 128 unsigned; 0x6400 | u = 1024 + u; -1152 makes it b exactly), pairs (b0, b2), (b1, b3)
__device__ __forceinline__ unsigned and_or_magic8(unsigned x, unsigned magic) {
  unsigned r;
  asm("v_and_or_b32 %0, %1, %2, %3" : "=v"(r) : "v"(x), "s"(0x00FF00FFu), "v"(magic));
  return r;
}
__device__ __forceinline__ f16x8 unpack_q8_0_f16(unsigned w0, unsigned w1, f16x2 d2, unsigned magic) {
  const f16x2 bias = {(_Float16)-1152.0f, (_Float16)-1152.0f};
  const unsigned x0 = w0 ^ 0x80808080u, x1 = w1 ^ 0x80808080u;
  const unsigned u0 = and_or_magic8(x0, magic), u1 = and_or_magic8(x0 >> 8, magic);
  const unsigned u2 = and_or_magic8(x1, magic), u3 = and_or_magic8(x1 >> 8, magic);
  const f16x2 p0 = (__builtin_bit_cast(f16x2, u0) + bias) * d2, p1 = (__builtin_bit_cast(f16x2, u1) + bias) * d2;
  const f16x2 p2 = (__builtin_bit_cast(f16x2, u2) + bias) * d2, p3 = (__builtin_bit_cast(f16x2, u3) + bias) * d2;
  return f16x8{p0[0], p0[1], p1[0], p1[1], p2[0], p2[1], p3[0], p3[1]};
}
// Q4_K: one class dword of the lane's pair; cs = {d sc_lo, -dmin m_lo, d sc_hi, -dmin m_hi}, each in both halves
struct Q4KF16Consts {
  f16x2 c_lo, o_lo, c_hi, o_hi;
};
__device__ __forceinline__ f16x8 unpack_q4_k_f16(unsigned w, const Q4KF16Consts& cs, unsigned magic) {
  const f16x2 bias = {(_Float16)-1024.0f, (_Float16)-1024.0f};
  const unsigned u0 = and_or_magic(w, magic), u1 = and_or_magic(w >> 8, magic);
  const unsigned u2 = and_or_magic(w >> 4, magic), u3 = and_or_magic(w >> 12, magic);
  const f16x2 p0 = __builtin_elementwise_fma(__builtin_bit_cast(f16x2, u0) + bias, cs.c_lo, cs.o_lo);
  const f16x2 p1 = __builtin_elementwise_fma(__builtin_bit_cast(f16x2, u1) + bias, cs.c_lo, cs.o_lo);
  const f16x2 p2 = __builtin_elementwise_fma(__builtin_bit_cast(f16x2, u2) + bias, cs.c_hi, cs.o_hi);
  const f16x2 p3 = __builtin_elementwise_fma(__builtin_bit_cast(f16x2, u3) + bias, cs.c_hi, cs.o_hi);
  return f16x8{p0[0], p0[1], p1[0], p1[1], p2[0], p2[1], p3[0], p3[1]};
}
__device__ __forceinline__ Q4KF16Consts q4k_f16_consts(i32x4 hdr, int g) {
  const unsigned f = q4k_pair_field((unsigned)hdr[1], (unsigned)hdr[2], (unsigned)hdr[3], g);
  const float d = h2f((unsigned short)((unsigned)hdr[0] & 0xffffu)), dmin = h2f((unsigned short)((unsigned)hdr[0] >> 16));
  const _Float16 c_lo = (_Float16)(d * (float)(f & 63u)), c_hi = (_Float16)(d * (float)((f >> 6) & 63u));  // (11 x 6 bits: exact in f32)
  const _Float16 o_lo = (_Float16)-(dmin * (float)((f >> 12) & 63u)), o_hi = (_Float16)-(dmin * (float)(f >> 18));
  return Q4KF16Consts{f16x2{c_lo, c_lo}, f16x2{o_lo, o_lo}, f16x2{c_hi, c_hi}, f16x2{o_hi, o_hi}};
}

// Q6_K: one dword of ql and the matching dword of qh, already shifted so that bits 0-1 of each byte belong to the low nibbles' quarter
// and bits 4-5 to the high nibbles'
__device__ __forceinline__ unsigned and_or_hi2(unsigned x, unsigned acc) {
  unsigned r;
  asm("v_and_or_b32 %0, %1, %2, %3" : "=v"(r) : "v"(x), "s"(0x00300030u), "v"(acc));
  return r;
}
__device__ __forceinline__ f16x8 unpack_q6_k_f16(unsigned w, unsigned hw, f16x2 c_lo, f16x2 c_hi, unsigned magic) {
  const f16x2 bias = {(_Float16)-1056.0f, (_Float16)-1056.0f};  // 1024 + q -> q - 32
  const unsigned u0 = and_or_hi2(hw << 4, and_or_magic(w, magic)), u1 = and_or_hi2(hw >> 4, and_or_magic(w >> 8, magic));
  const unsigned u2 = and_or_hi2(hw, and_or_magic(w >> 4, magic)), u3 = and_or_hi2(hw >> 8, and_or_magic(w >> 12, magic));
  const f16x2 p0 = (__builtin_bit_cast(f16x2, u0) + bias) * c_lo, p1 = (__builtin_bit_cast(f16x2, u1) + bias) * c_lo;
  const f16x2 p2 = (__builtin_bit_cast(f16x2, u2) + bias) * c_hi, p3 = (__builtin_bit_cast(f16x2, u3) + bias) * c_hi;
  return f16x8{p0[0], p0[1], p1[0], p1[1], p2[0], p2[1], p3[0], p3[1]};
}

enum { WF_Q4_0 = 0, WF_Q8_0 = 1, WF_Q4_K = 2, WF_Q6_K = 3, WF_Q4_1 = 4 };  // the weight format of a GEMM

template <int T_>
struct GemmF16Geo {
  static constexpr int T = T_, CW = 16 * T;    // column tiles per wave / prompt rows per workgroup
  static constexpr int KCH = 4;                // blocks per chunk
  // An LDS buffer is [block of the chunk g][column][80 bytes]: a column's 64 bytes of one block + 16 of padding.  ds_read_b128 is
  // serviced in four 16-lane groups that mix lanes of two k-slot groups g (MI355X_MICROARCH.md, LDS: {0-3, 12-15, 20-27}, ...) over 64
  // banks: with 80-byte columns the 16 columns of a tile sit on the 16 different 16-byte slots of the 256-byte bank row (5 i mod 16),
  // and the blocks' planes are a multiple of 256 bytes apart, so every lane group is conflict-free whichever g its lanes belong to.
  static constexpr int CSTR = 80, GSTR = CW * CSTR;
  static_assert(GSTR % 256 == 0, "block planes must be whole bank rows apart");
  static constexpr int BUF = KCH * GSTR, LDS_BYTES = 2 * BUF;
  static constexpr int B_LOADS = CW * KCH * 4 / 256;  // 16-byte pieces per thread and chunk (T)
};

// Up to three weight matrices against the same rhs in ONE launch (wq | wk | wv: the 1024-row k / v GEMMs alone leave most of the
// chip idle for a whole serial k loop): row tile rt belongs to the first matrix whose cumulative tile count exceeds it.
struct F16wMats {
  const i32x4* wq[3];  // the quant plane
  const char* wd[3];   // the scale plane (f16 per block; Q4_1: (d, m); Q4_K: the 16-byte headers; Q6_K: the qh plane)
  const char* ws2[3];  // Q6_K: the int8 scales (16 per super-block) ...
  const char* ws3[3];  // ... and d (f16 per super-block)
  float* out[3];
  float* out2[3];    // ksplit > 1: k piece s >= 1 writes its partial tiles to out2[j] + (s - 1) pstride (k_addn_f32 adds them)
  size_t pstride;
  const unsigned short* exp_tab;  // GU launches: the f16 exp table of silu (silu.rs:6-13)
  F16wHQuant hq;                   // GU launches: h leaves as Q8_0 / Q8_1 planes (+ ffn_down's B') instead of f32 (planes == nullptr: f32)
  int m[3];
  int tiles_end[3];  // cumulative row tiles
};
// h = silu(g) * u (silu.rs:6-13, arithmetic.rs:57-66; fused_ffn.hpp silu_mul: the exp through the reference's f16 table)
__device__ __forceinline__ float f16w_silu_mul(float g, float u, const unsigned short* __restrict__ exp_tab) {
  const float nexp = h2f(exp_tab[f2h(-g)]);
  return (g / (1.0f + nexp)) * u;
}
// GU (F = 2, two matrices of the same shape: ffn_gate and ffn_up): fragment 0 holds 16 rows of the FIRST matrix, fragment 1 the same
// 16 rows of the SECOND -- the workgroup owns 64 rows of both -- and the epilogue stores h = silu(g) * u (out[0]) instead of g and u:
// the (rows, hidden) f32 pair never makes its trip through memory (llama2.rs:620-630).  One k piece only (silu is not linear).
template <int WF, int F, int T_, bool GU = false>  // F 16-row fragments x T_ 16-column tiles per wave: the workgroup's four waves own 64 F consecutive weight rows
// ksplit > 1: that many workgroups per output tile, each over its piece of k, each writing its own partial buffer -- ffn_down and wo
// have few row tiles and a long serial k loop (one wave per SIMD otherwise), and a short pass has few tiles altogether
__global__ __launch_bounds__(256, 2) void k_gemm_f16w(F16wMats mats, const i32x4* __restrict__ xh, int nb, int n, int row_tiles, int ksplit) {
  using G = GemmF16Geo<T_>;
  constexpr int T = G::T, KCH = G::KCH;
  extern __shared__ __attribute__((aligned(16))) unsigned char f16w_lds[];
  const int tid = threadIdx.x, lane = tid & 63, wave = wave_in_wg();
  const int i = lane & 15, g = lane >> 4;
  int rt, ct, ks;
  {  // XCD-aware tile order (gemm_mfma.hip): the column tiles (and k halves) of a weight row tile back to back on ONE XCD
    const int per_row = (int)gridDim.x / row_tiles, col_tiles = per_row / ksplit;
    int sub;
    if ((row_tiles & 7) == 0) {
      const int x = (int)blockIdx.x & 7, j = (int)blockIdx.x >> 3;
      sub = j % per_row;
      rt = (j / per_row) * 8 + x;
    } else {
      rt = (int)blockIdx.x % row_tiles;
      sub = (int)blockIdx.x / row_tiles;
    }
    ct = sub % col_tiles;
    ks = sub / col_tiles;
  }
  static_assert(!GU || F == 2, "gate | up: one fragment of each matrix per wave");
  const int ti = GU ? 0 : rt < mats.tiles_end[0] ? 0 : rt < mats.tiles_end[1] ? 1 : 2;  // (uniform)
  // fragment f's matrix: the row tile's (ti), or -- GU -- matrix f
#define F16W_SEL(arr, j) ((j) == 0 ? mats.arr[0] : (j) == 1 ? mats.arr[1] : mats.arr[2])
#define F16W_PTRS(f)                                                                   \
  const int mj = GU ? (f) : ti;                                                        \
  const i32x4* __restrict__ wq = F16W_SEL(wq, mj);                                     \
  const char* __restrict__ wsc = F16W_SEL(wd, mj);                                     \
  const unsigned short* __restrict__ wd = (const unsigned short*)wsc;                  \
  const i32x4* __restrict__ wh = (const i32x4*)wsc;                                    \
  const i32x4* __restrict__ w6s = (const i32x4*)F16W_SEL(ws2, mj);                     \
  const unsigned short* __restrict__ w6d = (const unsigned short*)F16W_SEL(ws3, mj);   \
  (void)wq; (void)wd; (void)wh; (void)w6s; (void)w6d; (void)wsc
  float* __restrict__ out = ks == 0 ? (ti == 0 ? mats.out[0] : ti == 1 ? mats.out[1] : mats.out[2])
                                    : (ti == 0 ? mats.out2[0] : ti == 1 ? mats.out2[1] : mats.out2[2]) + (size_t)(ks - 1) * mats.pstride;
  const int m = ti == 0 ? mats.m[0] : ti == 1 ? mats.m[1] : mats.m[2];
  const int rt_l = rt - (ti == 0 ? 0 : ti == 1 ? mats.tiles_end[0] : mats.tiles_end[1]);
  constexpr int FSTEP = GU ? 0 : 16;  // rows between a wave's fragments
  const int r0 = GU ? rt_l * 64 + wave * 16 : rt_l * 64 * F + wave * 16 * F, c0 = ct * G::CW;
  // this workgroup's chunks: [ch_lo, ch_lo + nchunks) of the row's ceil(nb / KCH)
  // (Q4_K: an even chunk is half 0 of its super-block -- the pieces start on super-block boundaries)
  const int all_chunks = (nb + KCH - 1) / KCH;
  const int per_piece = (WF == WF_Q4_K || WF == WF_Q6_K) ? (((all_chunks + ksplit - 1) / ksplit + 1) & ~1) : (all_chunks + ksplit - 1) / ksplit;
  const int ch_lo = ks * per_piece;
  const int nchunks = all_chunks - ch_lo < per_piece ? all_chunks - ch_lo : per_piece;  // (>= 1: the launcher splits only long rows)

  // A: the lane's block (row i of fragment f, block kb0 + g) and its scale.  HBM latency is several chunk times (a chunk is ~0.4 us of
  // MFMAs and a workgroup has the SIMD almost to itself): a RING of four register sets, chunk c + 3 requested while chunk c is
  // multiplied; B' (L2-resident) two chunks ahead in two register sets.  All ring indices are compile-time (chunk loop unrolled by 4).
  constexpr int NQ = (WF == WF_Q8_0 || WF == WF_Q6_K) ? 2 : 1;  // 16-byte quant loads per fragment and chunk
  i32x4 aq[4][NQ * F];
  unsigned ad[4][F];   // Q4_0 / Q8_0: the block's scale; Q4_1: d | m << 16
  i32x4 hq[2][F];      // Q4_K: the super-block header of ring slots (0, 1) / (2, 3), fetched with the even slot; Q6_K: the 16 scales
  unsigned hd[2][F];   // Q6_K: d
  auto fetch_a = [&](auto Jc, int ch) {
    constexpr int J = decltype(Jc)::value;
    const int cc = ch_lo + (ch < nchunks ? ch : nchunks - 1);  // (past the end: re-read the last chunk, never consumed)
    if constexpr (WF == WF_Q4_K) {
      const int nsb = nb >> 3, sb = cc >> 1, h = cc & 1;
#pragma unroll
      for (int f = 0; f < F; f++) {
        F16W_PTRS(f);
        const int row = r0 + FSTEP * f + i;
        const size_t blk = (size_t)(row < m ? row : m - 1) * nsb + sb;
        aq[J][f] = __builtin_nontemporal_load(wq + blk * 8 + 2 * g + h);
        if constexpr ((J & 1) == 0) hq[J >> 1][f] = __builtin_nontemporal_load(wh + blk);
      }
    } else if constexpr (WF == WF_Q6_K) {
      const int nsb = nb >> 3, sb = cc >> 1, half = cc & 1;
#pragma unroll
      for (int f = 0; f < F; f++) {
        F16W_PTRS(f);
        const int row = r0 + FSTEP * f + i;
        const size_t blk = (size_t)(row < m ? row : m - 1) * nsb + sb;
        aq[J][2 * f] = __builtin_nontemporal_load(wq + blk * 8 + 4 * half + g);
        aq[J][2 * f + 1] = __builtin_nontemporal_load(wh + blk * 4 + 2 * half + (g & 1));  // (wh: the qh plane, 64 bytes per super-block)
        if constexpr ((J & 1) == 0) {
          hq[J >> 1][f] = __builtin_nontemporal_load(w6s + blk);
          hd[J >> 1][f] = __builtin_nontemporal_load(w6d + blk);
        }
      }
    } else {
      const int kb = cc * KCH + g;
      const int gkb = kb < nb ? kb : nb - 1;
#pragma unroll
      for (int f = 0; f < F; f++) {
        F16W_PTRS(f);
        const int row = r0 + FSTEP * f + i;
        const size_t blk = (size_t)(row < m ? row : m - 1) * nb + gkb;
#pragma unroll
        for (int u = 0; u < NQ; u++) aq[J][NQ * f + u] = __builtin_nontemporal_load(wq + blk * NQ + u);
        unsigned dv;
        if constexpr (WF == WF_Q4_1)
          dv = __builtin_nontemporal_load((const unsigned*)wsc + blk);
        else
          dv = __builtin_nontemporal_load(wd + blk);
        ad[J][f] = kb < nb ? dv : 0u;  // past the row's end: scale 0 (and m 0), the slots add nothing
      }
    }
  };
  // B': 128 columns x 256 bytes per chunk, 8 pieces per thread (piece p: column p / 16, 16 bytes p % 16 of the chunk): ONE 32-bit
  // byte offset per thread, advanced by a chunk = 256 bytes per fetch (the fetches come in chunk order), against a scalar base per
  // piece -- 16 columns further each (global_load ... s[base], v offset: no per-piece pointer registers).  Columns past n (a ragged
  // last tile) and the look-ahead of the last iterations read whatever follows in xh -- launch_gemm_f16w's contract keeps that
  // inside the allocation; finite or not, it meets only output columns that are never stored, zero weights, or is never consumed.
  // register sets of B' pieces in flight (one wave per SIMD: the longer look-ahead; the 32-byte-per-lane formats with two fragments:
  // one set -- B' is L2-resident and a chunk of 64 MFMAs per wave is longer than an L2 round trip)
  constexpr int NBD = (F == 1 && WF == WF_Q4_0) ? 4 : (F == 2 && (WF == WF_Q8_0 || WF == WF_Q6_K)) ? 1 : 2;
  i32x4 rb[NBD][G::B_LOADS];
  const char* xbase = (const char*)xh + ((size_t)c0 * nb + (size_t)ch_lo * KCH) * 64;  // (uniform)
  const unsigned xstep = 16u * (unsigned)nb * 64u;                                        // 16 columns
  unsigned xoff = (unsigned)(tid >> 4) * (unsigned)nb * 64u + (unsigned)(tid & 15) * 16u;
  auto fetch_b = [&](i32x4 (&r)[G::B_LOADS], int ch) {
    (void)ch;
#pragma unroll
    for (int u = 0; u < G::B_LOADS; u++) r[u] = *(const i32x4*)(xbase + (size_t)u * xstep + xoff);
    xoff += KCH * 64;
  };
  auto commit_b = [&](const i32x4 (&r)[G::B_LOADS], int buf) {
    unsigned char* S = f16w_lds + buf * G::BUF;
#pragma unroll
    for (int u = 0; u < G::B_LOADS; u++) {
      const int p = tid + 256 * u, col = p >> 4, pc = p & 15;
      *(i32x4*)(S + (pc >> 2) * G::GSTR + col * G::CSTR + (pc & 3) * 16) = r[u];
    }
  };

  f32x4 acc[F][T];
#pragma unroll
  for (int f = 0; f < F; f++)
#pragma unroll
    for (int t = 0; t < T; t++) acc[f][t] = f32x4{0.f, 0.f, 0.f, 0.f};
  unsigned magic = 0x64006400u;
  asm volatile("" : "+v"(magic));  // pinned in a VGPR (and_or_magic)

  fetch_b(rb[0], 0);
  fetch_a(std::integral_constant<int, 0>{}, 0);
  fetch_a(std::integral_constant<int, 1>{}, 1);
  fetch_a(std::integral_constant<int, 2>{}, 2);
  Q4KF16Consts cs[F];  // Q4_K: the lane's pair's constants of the current super-block (made in the even slot); Q4_1: (d, m); Q6_K: d sc
  commit_b(rb[0], 0);
  // chunk c sits in rb[c % NBD] from NBD chunks before it is committed (the fetches come in chunk order: pb advances)
#pragma unroll
  for (int c = 1; c <= NBD; c++) fetch_b(rb[c % NBD], c);
  __syncthreads();
  // chunk ch (ring slot J, LDS buffer ch & 1)
  auto chunk = [&](auto Jc, int ch) {
    constexpr int J = decltype(Jc)::value;
    fetch_a(std::integral_constant<int, (J + 3) & 3>{}, ch + 3);
    const unsigned char* S = f16w_lds + (J & 1) * G::BUF + g * G::GSTR + i * G::CSTR;
    f16x2 d2[F];
#pragma unroll
    for (int f = 0; f < F; f++) {
      if constexpr (WF == WF_Q4_K) {
        if constexpr ((J & 1) == 0) cs[f] = q4k_f16_consts(hq[J >> 1][f], g);
      } else if constexpr (WF == WF_Q6_K) {
        // the lane's two scales of a half: sc[8 half + (g & 1) + 2 (g >> 1)] (low nibbles' quarter) and 4 further (high nibbles');
        // both halves' in the even slot (the odd slot's fetch re-uses the header registers): c_* = half 0, o_* = half 1
        if constexpr ((J & 1) == 0) {
          const int sh = 8 * ((g & 1) + 2 * (g >> 1));
          const float d = h2f((unsigned short)hd[J >> 1][f]);
          _Float16 c[4];
#pragma unroll
          for (int u = 0; u < 4; u++) c[u] = (_Float16)(d * (float)(signed char)(((unsigned)hq[J >> 1][f][u] >> sh) & 0xffu));
          cs[f].c_lo = f16x2{c[0], c[0]};
          cs[f].c_hi = f16x2{c[1], c[1]};
          cs[f].o_lo = f16x2{c[2], c[2]};
          cs[f].o_hi = f16x2{c[3], c[3]};
        }
      } else if constexpr (WF == WF_Q4_1) {
        const unsigned dm = ad[J][f];
        cs[f].c_lo = cs[f].c_hi = __builtin_bit_cast(f16x2, (dm & 0xffffu) | (dm << 16));
        cs[f].o_lo = cs[f].o_hi = __builtin_bit_cast(f16x2, (dm >> 16) | (dm & 0xffff0000u));
      } else {
        d2[f] = __builtin_bit_cast(f16x2, ad[J][f] | (ad[J][f] << 16));
      }
    }
    // B' fragments in groups of TG tiles, two groups in registers: group q + 1 is read from LDS while group q is multiplied (left
    // to itself the compiler reads ONE fragment, waits, multiplies: 32 exposed LDS round trips per chunk with the SIMD almost to
    // itself).  The scheduling barriers keep the reads of a group ahead of the MFMAs of the previous one.
    constexpr int TG = T >= 4 ? 4 : T, NG = 4 * (T / TG);  // groups per chunk (4 steps x T / TG)
    f16x8 bq[2][TG];
    auto read_group = [&](f16x8 (&dst)[TG], int q) {
      const int s = q / (T / TG), t0 = (q % (T / TG)) * TG;
#pragma unroll
      for (int t = 0; t < TG; t++) dst[t] = *(const f16x8*)(S + (t0 + t) * 16 * G::CSTR + s * 16);
    };
    read_group(bq[0], 0);
    f16x8 a[F];
#pragma unroll
    for (int q = 0; q < NG; q++) {
      const int s = q / (T / TG), t0 = (q % (T / TG)) * TG;
      if (q + 1 < NG) read_group(bq[(q + 1) & 1], q + 1);
      if (q % (T / TG) == 0) {
#pragma unroll
        for (int f = 0; f < F; f++) {
          if constexpr (WF == WF_Q4_0) a[f] = unpack_q4_0_f16((unsigned)aq[J][f][s], d2[f], magic);
          if constexpr (WF == WF_Q8_0) a[f] = unpack_q8_0_f16((unsigned)aq[J][2 * f][s], (unsigned)aq[J][2 * f + 1][s], d2[f], magic);
          if constexpr (WF == WF_Q4_K || WF == WF_Q4_1) a[f] = unpack_q4_k_f16((unsigned)aq[J][f][s], cs[f], magic);
          if constexpr (WF == WF_Q6_K)
            a[f] = unpack_q6_k_f16((unsigned)aq[J][2 * f][s], (unsigned)aq[J][2 * f + 1][s] >> (2 * (g >> 1)), (J & 1) ? cs[f].o_lo : cs[f].c_lo,
                                   (J & 1) ? cs[f].o_hi : cs[f].c_hi, magic);
        }
      }
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int t = 0; t < TG; t++)
#pragma unroll
        for (int f = 0; f < F; f++) acc[f][t0 + t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[f], bq[q & 1][t], acc[f][t0 + t], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
    }
    commit_b(rb[(J + 1) % NBD], (J + 1) & 1);  // chunk ch + 1 into the other buffer (read last in iteration ch - 1: behind its barrier)
    fetch_b(rb[(J + 1) % NBD], ch + 1 + NBD);  // ... and the freed register set takes chunk ch + 1 + NBD
    __syncthreads();
  };
  for (int ch = 0; ch < nchunks; ch += 4) {  // (uniform conditions: every thread takes the barrier inside a chunk or none does)
    chunk(std::integral_constant<int, 0>{}, ch);
    if (ch + 1 < nchunks) chunk(std::integral_constant<int, 1>{}, ch + 1);
    if (ch + 2 < nchunks) chunk(std::integral_constant<int, 2>{}, ch + 2);
    if (ch + 3 < nchunks) chunk(std::integral_constant<int, 3>{}, ch + 3);
  }
  // D: lane (i, g) holds rows 4 g .. 4 g + 3 of column i of every tile
  if constexpr (GU) {
    if (mats.hq.planes != nullptr) {
      // h = silu(g) * u of the workgroup's 64 rows x CW columns through LDS (the B' buffers are done), then ONE THREAD per
      // (column, 32-row block) runs the row quantizer on it (quant_lane32's arithmetic, buf_q8_0.rs:87-134 / buf_q8_1.rs:90-129: the
      // block maximum and the integer sum do not depend on the order) and writes the block's quants, scale, sum and -- in the slot
      // order of ffn_down's GEMM -- its 32 pre-scaled halfs: h never exists as f32 in memory
      constexpr int CS = 68;  // floats per column (64 rows + 4: 16-byte aligned columns, four banks apart)
      float* H = (float*)f16w_lds;
      __syncthreads();
#pragma unroll
      for (int t = 0; t < T; t++) {
        f32x4 hv;
#pragma unroll
        for (int r = 0; r < 4; r++) hv[r] = f16w_silu_mul(acc[0][t][r], acc[1][t][r], mats.exp_tab);
        *(f32x4*)(H + (16 * t + i) * CS + wave * 16 + 4 * g) = hv;
      }
      __syncthreads();
      const F16wHQuant& hq = mats.hq;
      const int nbh = m / 32;
      for (int b = tid; b < G::CW * 2; b += 256) {
        const int col = c0 + (b >> 1), hb = rt_l * 2 + (b & 1);
        if (col >= n || hb >= nbh) continue;
        f32x4 v[8];
#pragma unroll
        for (int j = 0; j < 8; j++) v[j] = *(const f32x4*)(H + (b >> 1) * CS + 32 * (b & 1) + 4 * j);
        float amax = 0.0f;
#pragma unroll
        for (int j = 0; j < 8; j++)
#pragma unroll
          for (int r = 0; r < 4; r++) amax = fmaxf(amax, fabsf(v[j][r]));
        const float dd = amax / 127.0f;
        const unsigned short dh = f2h(dd);
        int q[32], sum = 0;
#pragma unroll
        for (int e = 0; e < 32; e++) {
          const float x = v[e >> 2][e & 3];
          if (hq.q81) {
            q[e] = (int)fminf(fmaxf(x / dd, -128.0f), 127.0f);
          } else {
            q[e] = (int)(signed char)(unsigned char)((unsigned)rs_f32_as_i32(x / dd) & 0xffu);
          }
          sum += q[e];
        }
        char* p = hq.planes + (size_t)col * hq.stride;
        i32x4 pk[2];
#pragma unroll
        for (int e = 0; e < 8; e++)
          pk[e >> 2][e & 3] = (int)(((unsigned)q[4 * e] & 0xffu) | (((unsigned)q[4 * e + 1] & 0xffu) << 8) | (((unsigned)q[4 * e + 2] & 0xffu) << 16) |
                                    (((unsigned)q[4 * e + 3] & 0xffu) << 24));
        ((i32x4*)(p + (size_t)hb * 32))[0] = pk[0];
        ((i32x4*)(p + (size_t)hb * 32))[1] = pk[1];
        ((unsigned short*)(p + hq.off_d))[hb] = dh;
        if (hq.q81)
          ((unsigned short*)(p + hq.off_aux))[hb] = f2h((float)sum * dd);
        else
          ((int*)(p + hq.off_aux))[hb] = sum;
        if (hq.xh != nullptr) {
          const float ds = h2f(dh);
          unsigned short* xr = hq.xh + ((size_t)col * nbh + hb) * 32;
#pragma unroll
          for (int s4 = 0; s4 < 4; s4++) {
            unsigned short o[8];
#pragma unroll
            for (int e = 0; e < 8; e++) o[e] = f16w_value(q[f16w_slot_elem(8 * s4 + e)], ds);
            *(i32x4*)(xr + 8 * s4) = i32x4{(int)(o[0] | ((unsigned)o[1] << 16)), (int)(o[2] | ((unsigned)o[3] << 16)),
                                           (int)(o[4] | ((unsigned)o[5] << 16)), (int)(o[6] | ((unsigned)o[7] << 16))};
          }
        }
      }
      return;
    }
#pragma unroll
    for (int t = 0; t < T; t++) {
      const int col = c0 + 16 * t + i;
      if (col >= n) continue;
      const int row = r0 + 4 * g;
      f32x4 hv;
#pragma unroll
      for (int r = 0; r < 4; r++) hv[r] = f16w_silu_mul(acc[0][t][r], acc[1][t][r], mats.exp_tab);
      float* o = out + (size_t)col * m + row;
      if (row + 3 < m) {
        *(f32x4*)o = hv;
      } else {
#pragma unroll
        for (int r = 0; r < 4; r++)
          if (row + r < m) o[r] = hv[r];
      }
    }
    return;
  }
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

#undef F16W_PTRS
#undef F16W_SEL

// hipFuncAttributeMaxDynamicSharedMemorySize, once per (device, kernel)
static bool f16w_raise_lds(const crabml_hip_device* dev, const void* fn, int bytes) {
  static std::mutex mu;
  static std::map<std::pair<int, const void*>, int> have;
  std::lock_guard<std::mutex> g(mu);
  int& cur = have[{dev->ordinal, fn}];
  if (bytes <= cur) return true;
  if (hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, bytes) != hipSuccess) return false;
  cur = bytes;
  return true;
}
// xh: the rows' pre-scaled f16 planes (launch_rows_to_f16) inside an allocation of gemm_f16w_xh_bytes(b, k): the kernel reads whole
// 128-column tiles and up to three chunks past the last column's end; returns false when the shape is not covered
template <int WF, int F, int T, bool GU = false>
static bool launch_f16w_t(crabml_hip_device* dev, const F16wMats& mats, int row_tiles, size_t k, const void* xh, size_t b, int ksplit) {
  using G = GemmF16Geo<T>;
  if (!f16w_raise_lds(dev, (const void*)k_gemm_f16w<WF, F, T, GU>, G::LDS_BYTES)) {  // (80 KB of dynamic LDS: raised once per device)
    (void)hipGetLastError();
    return false;
  }
  const int col_tiles = (int)((b + G::CW - 1) / G::CW);
  k_gemm_f16w<WF, F, T, GU><<<dim3(row_tiles * col_tiles * ksplit), 256, G::LDS_BYTES, dev->stream>>>(mats, (const i32x4*)xh, (int)(k / 32), (int)b,
                                                                                          row_tiles, ksplit);
  return true;
}
size_t gemm_f16w_xh_bytes(size_t rows, size_t k) { return ((rows + 127) / 128 * 128) * k * 2 + 4096; }
// the weight formats the kernel covers, and the rows' format each pairs with (CpuTensorBuf::quantize's choice: buf/api.rs:142-159)
bool gemm_f16w_covers(uint32_t w_dtype, uint32_t act_qtype) {
  if (act_qtype == CRABML_HIP_Q8_0) return w_dtype == CRABML_HIP_Q4_0 || w_dtype == CRABML_HIP_Q8_0;
  if (act_qtype == CRABML_HIP_Q8_1) return w_dtype == CRABML_HIP_Q4_1;
  return act_qtype == CRABML_HIP_Q8_K && (w_dtype == CRABML_HIP_Q4_K || w_dtype == CRABML_HIP_Q6_K);
}
// wave tile: T column tiles of 16 prompt rows -- 8 (128-row workgroup tiles) from 65 rows up; short passes (a short prompt, the ragged
// tail of a long one) take 4 / 2: the MFMAs of an empty column tile cost what a full one's do
static int f16w_col_tiles_per_wave(size_t b, int variant) { return (variant & 512) ? 4 : (variant & 16) ? 8 : b <= 32 ? 2 : b <= 64 ? 4 : 8; }
template <int WF>
static bool launch_f16w_fmt(crabml_hip_device* dev, const F16wMats& mats, int row_tiles, size_t k, const void* xh, size_t b, int ksplit, int F,
                            int variant, bool gu, int T) {
  if (gu) {
    switch (T) {
      case 2: return launch_f16w_t<WF, 2, 2, true>(dev, mats, row_tiles, k, xh, b, 1);
      case 4: return launch_f16w_t<WF, 2, 4, true>(dev, mats, row_tiles, k, xh, b, 1);
      default: return launch_f16w_t<WF, 2, 8, true>(dev, mats, row_tiles, k, xh, b, 1);
    }
  }
  switch (T) {
    case 2: return F == 2 ? launch_f16w_t<WF, 2, 2>(dev, mats, row_tiles, k, xh, b, ksplit) : launch_f16w_t<WF, 1, 2>(dev, mats, row_tiles, k, xh, b, ksplit);
    case 4: return F == 2 ? launch_f16w_t<WF, 2, 4>(dev, mats, row_tiles, k, xh, b, ksplit) : launch_f16w_t<WF, 1, 4>(dev, mats, row_tiles, k, xh, b, ksplit);
    default: return F == 2 ? launch_f16w_t<WF, 2, 8>(dev, mats, row_tiles, k, xh, b, ksplit) : launch_f16w_t<WF, 1, 8>(dev, mats, row_tiles, k, xh, b, ksplit);
  }
}
// nw weight matrices (one format, the same k) against the same rhs rows: out[j] (b, m[j]) = W[j] . x
// ws (nullable), ws_floats: scratch for split launches.  When the output tiles alone leave the chip underfilled -- wo / ffn_down (few
// row tiles, a long k), any short pass -- the k range is cut into 2 / 4 / 8 pieces, one workgroup each: piece 0 writes out, the
// others their own partial buffers in ws, and k_addn_f32 adds them in piece order.  (Measured and not kept: the pieces added with
// f32 atomics onto a zeroed output -- 32.4k -> 30.1k prompt tok/s, and the sum's order would vary from run to run.)
bool launch_gemm_f16w(crabml_hip_device* dev, const crabml_hip_buf* const* w, const size_t* m, int nw, size_t k, const void* xh, size_t b,
                      float* const* out, float* ws, size_t ws_floats, const unsigned short* gu_exp_tab, int* gu_done, int* defer_parts,
                      const F16wHQuant* hq) {
  if (gu_done) *gu_done = 0;
  if (defer_parts) *defer_parts = 0;
  if (nw < 1 || nw > 3 || k % 32 != 0 || b < 16) return false;
  const uint32_t dt = w[0]->dtype;
  if (dt != CRABML_HIP_Q4_0 && dt != CRABML_HIP_Q8_0 && dt != CRABML_HIP_Q4_K && dt != CRABML_HIP_Q6_K && dt != CRABML_HIP_Q4_1) return false;
  if ((dt == CRABML_HIP_Q4_K || dt == CRABML_HIP_Q6_K) && k % 256 != 0) return false;
  for (int j = 0; j < nw; j++)
    if (w[j]->dtype != dt || m[j] % 4 != 0) return false;
  static const int variant = [] {  // lab hook (CRABML_HIP_TEST_HOOKS=1 CRABML_HIP_F16W=n): 1 = two fragments, 3 = one; +8 = never split k; +16 = T = 8 always; +32 = k pieces of >= 8 chunks; +64 = no gate | up epilogue; +512 = T = 4 always; +1024 = narrow launches as <2, 4>
    const char* h = getenv("CRABML_HIP_TEST_HOOKS");
    const char* e = getenv("CRABML_HIP_F16W");
    return h && h[0] == '1' && e ? atoi(e) : 0;
  }();
  size_t mtot = 0;
  for (int j = 0; j < nw; j++) mtot += m[j];
  int T = f16w_col_tiles_per_wave(b, variant);
  size_t cw = 16 * (size_t)T, col128 = (b + cw - 1) / cw;  // (column tiles of the launch)
  // two fragments per wave (every B' fragment read from LDS feeds two MFMAs) when 128-row tiles still cover the chip
  int F = ((mtot + 127) / 128) * col128 >= (size_t)dev->n_cu ? 2 : 1;
  if ((variant & 1024) && F == 1 && T == 8 && b >= 128) {  // lab: narrow launches as two fragments x four column tiles
    F = 2;
    T = 4;
    cw = 64;
    col128 = (b + cw - 1) / cw;
  }
  if ((variant & 7) == 1) F = 2;
  if ((variant & 7) == 3) F = 1;
  // gate | up with the SiLU * mul epilogue (out[0] = h, out[1] untouched): when 64-row tiles of both matrices cover the chip without
  // cutting k
  const bool gu = gu_exp_tab != nullptr && gu_done != nullptr && nw == 2 && m[0] == m[1] && !(variant & 64) &&
                  ((m[0] + 63) / 64) * col128 * 2 >= (size_t)dev->n_cu * 3;
  if (gu) F = 2;
  F16wMats mats{};
  mats.exp_tab = gu_exp_tab;
  const bool hquant = gu && hq != nullptr && hq->planes != nullptr && m[0] % 32 == 0;
  if (hquant) mats.hq = *hq;
  F16wParts parts{};
  int row_tiles = 0;
  size_t before = 0;  // output elements of the matrices before j
  for (int j = 0; j < 3; j++) {
    const int jj = j < nw ? j : nw - 1;
    const char* wp = (const char*)w[jj]->ptr;
    mats.wq[j] = (const i32x4*)wp;
    mats.wd[j] = wp + w[jj]->wl.off_scale;
    mats.ws2[j] = wp + w[jj]->wl.off_scale + w[jj]->wl.n_blocks * 64;  // (Q6_K: ql | qh | scales | d, common.hpp)
    mats.ws3[j] = wp + w[jj]->wl.off_scale + w[jj]->wl.n_blocks * 80;
    mats.out[j] = out[jj];
    mats.out2[j] = ws + before;
    mats.m[j] = (int)m[jj];
    parts.out[j] = out[jj];
    parts.part[j] = ws + before;
    if (j < nw) {
      if (!gu) row_tiles += (int)((m[j] + 64 * F - 1) / (64 * F));
      before += b * m[j];
    }
    if (gu) row_tiles = (int)((m[0] + 63) / 64);
    mats.tiles_end[j] = row_tiles;
    parts.end4[j] = before / 4;
  }
  mats.pstride = parts.pstride = before;
  // pieces of k: double while the launch is short of ~1.5 workgroups per CU, every piece keeps >= 4 chunks (and whole super-blocks),
  // and the partial buffers fit the scratch
  int ksplit = 1;
  const size_t chunks = (k + 127) / 128;
  if (ws != nullptr && !(variant & 8) && k % 128 == 0 && !gu)
    while (ksplit < 8 && (size_t)row_tiles * col128 * ksplit < (size_t)dev->n_cu * 3 / 2 && chunks % (size_t)(4 * ksplit) == 0 &&
           chunks / (size_t)(2 * ksplit) >= (size_t)((variant & 32) ? 8 : 4) && (size_t)(2 * ksplit - 1) * before <= ws_floats)
      ksplit *= 2;
  bool ok;
  if (dt == CRABML_HIP_Q8_0)
    ok = launch_f16w_fmt<WF_Q8_0>(dev, mats, row_tiles, k, xh, b, ksplit, F, variant, gu, T);
  else if (dt == CRABML_HIP_Q4_K)
    ok = launch_f16w_fmt<WF_Q4_K>(dev, mats, row_tiles, k, xh, b, ksplit, F, variant, gu, T);
  else if (dt == CRABML_HIP_Q6_K)
    ok = launch_f16w_fmt<WF_Q6_K>(dev, mats, row_tiles, k, xh, b, ksplit, F, variant, gu, T);
  else if (dt == CRABML_HIP_Q4_1)
    ok = launch_f16w_fmt<WF_Q4_1>(dev, mats, row_tiles, k, xh, b, ksplit, F, variant, gu, T);
  else
    ok = launch_f16w_fmt<WF_Q4_0>(dev, mats, row_tiles, k, xh, b, ksplit, F, variant, gu, T);
  if (ok && gu) *gu_done = hquant ? 2 : 1;
  if (ok && ksplit > 1 && defer_parts != nullptr && nw == 1) {
    *defer_parts = ksplit - 1;  // the caller's next row kernel adds the pieces (ws + s * b * m, s = 0 ..) in the same order
  } else if (ok && ksplit > 1) {
    parts.nparts = ksplit - 1;
    k_addn_f32<<<(unsigned)((parts.end4[2] + 255) / 256), 256, 0, dev->stream>>>(parts);
  }
  return ok;
}

}  // namespace crabml_hip

// gemv.hip -- block-quantized decode GEMV for gfx950 (the kernel the HBM-roofline target is on).
//
// Replaces gemv_dense_2d_2d + CpuTensorBuf::vec_dot (crabml-core/src/cpu/primitives/matmul_vec.rs:26-78,
// crabml-core/src/cpu/buf/api.rs:230-249) and the per-format vec_dot_* kernels:
//   Q4_0 x Q8_0  buf_q4_0.rs:240-253      Q8_0 x Q8_0  buf_q8_0.rs:275-286
//   Q4_1 x Q8_1  buf_q4_1.rs:266-280      Q4_K x Q8_K  buf_q4_k.rs:192-277      Q8_K x Q8_K  buf_q8_k.rs:211-224
//   Q6_K x Q8_K  buf_q6_k.rs:183-234
//   F32 x F32    buf_f32.rs:19-27         F16 x F16    buf_f16.rs:83-97
//
// Mapping (chosen by measurement, profiles/r01_gemv_lab_layout_sweep.log): one wavefront owns R
// consecutive output rows; lane l owns quant blocks l, l+64, ... of each row, so one wave step is a
// single aligned 1 KiB `global_load_dwordx4 ... nt` of packed nibbles plus a 128-byte load of the
// f16 scales.  The integer part of every block product (nibble unpack, -8 offset, int8 dot) is exact
// (v_dot4_i32_i8); the per-block f32 scaling follows the reference's expression
// `(sumi as f32 * d_w) * d_x`; only the ORDER in which block terms are added differs from the scalar
// CPU loop (lane-strided partial sums + a 64-lane butterfly), which is why logits carry an fp tolerance.
// The kernel is HBM-bound (~3.6 flop/byte): no LDS round trip, no MFMA; x is re-read from L1/L2.
#include "gemv_core.hpp"
#include "kernels.hpp"

namespace crabml_hip {

// ---- Q4_0 x Q8_0 ---------------------------------------------------------------------------------
template <int R>
__global__ __launch_bounds__(256) void k_gemv_q4_0(const i32x4* __restrict__ wq, const unsigned short* __restrict__ wd,
                                                   ActQ8_0 act, float* __restrict__ out, int m, int nb) {
  const int lane = threadIdx.x & 63;
  const int wave = blockIdx.x * (blockDim.x >> 6) + wave_in_wg();
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
      q[r] = __builtin_nontemporal_load(wq + idx);
      dw[r] = __builtin_nontemporal_load(wd + idx);
    }
    i32x4 xlo = act.q[2 * b], xhi = act.q[2 * b + 1];
    float dx = h2f(act.d[b]);
    int xs = act.isum[b];
#pragma unroll
    for (int r = 0; r < R; r++) {
      int si = dot_q4_0(q[r], xlo, xhi, xs);
      acc[r] += ((float)si * h2f(dw[r])) * dx;  // buf_q4_0.rs:249
    }
  }
#pragma unroll
  for (int r = 0; r < R; r++) {
    float s = wave_sum_f32(acc[r]);
    if (lane == 0 && row0 + r < m) out[row0 + r] = s;
  }
}

// ---- Q8_0 x Q8_0 ---------------------------------------------------------------------------------
template <int R>
__global__ __launch_bounds__(256) void k_gemv_q8_0(const i32x4* __restrict__ wq, const unsigned short* __restrict__ wd,
                                                   ActQ8_0 act, float* __restrict__ out, int m, int nb) {
  const int lane = threadIdx.x & 63;
  const int wave = blockIdx.x * (blockDim.x >> 6) + wave_in_wg();
  const int row0 = wave * R;
  if (row0 >= m) return;
  float acc[R];
  rows_partial<CRABML_HIP_Q8_0, R>(wq, wd, act, row0, m, nb, lane, acc);  // half a block per lane (gemv_core.hpp)
#pragma unroll
  for (int r = 0; r < R; r++) {
    float s = wave_sum_f32(acc[r]);
    if (lane == 0 && row0 + r < m) out[row0 + r] = s;
  }
}

// ---- Q4_1 x Q8_1 ---------------------------------------------------------------------------------
template <int R>
__global__ __launch_bounds__(256) void k_gemv_q4_1(const i32x4* __restrict__ wq, const unsigned* __restrict__ wdm,
                                                   ActQ8_1 act, float* __restrict__ out, int m, int nb) {
  const int lane = threadIdx.x & 63;
  const int wave = blockIdx.x * (blockDim.x >> 6) + wave_in_wg();
  const int row0 = wave * R;
  if (row0 >= m) return;
  float acc[R];
  rows_partial<CRABML_HIP_Q4_1, R>(wq, (const unsigned short*)wdm, act, row0, m, nb, lane, acc);  // gemv_core.hpp
#pragma unroll
  for (int r = 0; r < R; r++) {
    float s = wave_sum_f32(acc[r]);
    if (lane == 0 && row0 + r < m) out[row0 + r] = s;
  }
}

// ---- Q4_K x Q8_K ---------------------------------------------------------------------------------
// planes: qs[n][128] | hdr[n][16] (d, dmin, scales[12]).  A lane owns one 16-byte qs chunk j of a super-block
// (8 lanes per super-block, so a wave's load is one aligned 1 KiB request, like the Q4_0 kernel): chunk j belongs
// to the 64-element pair p = j / 2 and carries, for positions 16 (j & 1) .. +16, the low nibbles of sub-block 2p
// and the high nibbles of sub-block 2p + 1 (buf_q4_k.rs:212-217).  The 16-byte header is fetched by all 8 lanes
// (one 128-byte request per wave) and the 6-bit (scale, min) pairs are unpacked in registers.  (Widening the
// 6-bit fields to bytes at upload -- 20-byte headers, two dword loads, 20 fewer VALU ops -- measured SLOWER:
// gate/up 8.98 -> 10.88 us; the loop is bound by memory instructions, not by VALU.)
template <int R>
__global__ __launch_bounds__(256) void k_gemv_q4_k(const i32x4* __restrict__ wq, const i32x4* __restrict__ wh, ActQ8_K act,
                                                   float* __restrict__ out, int m, int nsb) {
  const int lane = threadIdx.x & 63;
  const int wave = blockIdx.x * (blockDim.x >> 6) + wave_in_wg();
  const int row0 = wave * R;
  if (row0 >= m) return;
  float acc[R];
  rows_partial_q4k<R>(wq, wh, act, row0, m, nsb, lane, acc);  // gemv_core.hpp
#pragma unroll
  for (int r = 0; r < R; r++) {
    float s = wave_sum_f32(acc[r]);
    if (lane == 0 && row0 + r < m) out[row0 + r] = s;
  }
}

// ---- Q5_K x Q8_K (buf_q5_k.rs:229-325; rows_partial_q5k, gemv_core.hpp) ---------------------------------------------------------
template <int R>
__global__ __launch_bounds__(256) void k_gemv_q5_k(const char* __restrict__ w, size_t off_qh, ActQ8_K act, float* __restrict__ out, int m,
                                                   int nsb) {
  const int lane = threadIdx.x & 63;
  const int wave = blockIdx.x * (blockDim.x >> 6) + wave_in_wg();
  const int row0 = wave * R;
  if (row0 >= m) return;
  float acc[R];
  rows_partial_q5k<R>(w, off_qh, act, row0, m, nsb, lane, acc);
#pragma unroll
  for (int r = 0; r < R; r++) {
    float s = wave_sum_f32(acc[r]);
    if (lane == 0 && row0 + r < m) out[row0 + r] = s;
  }
}

// ---- Q5_0 / Q5_1 / Q2_K / Q3_K (piece policies, gemv_core.hpp) -------------------------------------------------------------
template <class P, int R>
__global__ __launch_bounds__(256) void k_gemv_pieces(const char* __restrict__ w, size_t off, size_t n, typename P::Act act,
                                                     float* __restrict__ out, int m, int nbr) {
  const int lane = threadIdx.x & 63;
  const int wave = blockIdx.x * (blockDim.x >> 6) + wave_in_wg();
  const int row0 = wave * R;
  if (row0 >= m) return;
  float acc[R];
  rows_partial_pieces<P, R>(w, off, n, act, row0, m, nbr, lane, acc);
#pragma unroll
  for (int r = 0; r < R; r++) {
    float s = wave_sum_f32(acc[r]);
    if (lane == 0 && row0 + r < m) out[row0 + r] = s;
  }
}
// parity hook: the exact integers of those formats through the production unpack -- one per 32-element block (Q5_0, Q5_1) or
// per 16-element scale group in element order (Q2_K, Q3_K)
template <class P>
__global__ void k_block_dots_pieces(const char* __restrict__ w, size_t off, size_t n, typename P::Act act, size_t blk0, int np,
                                    int* __restrict__ out) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= np) return;
  const typename P::W wv = P::load(w, off, n, blk0, c);
  const typename P::X x = P::loadx(act, c);
  int o[P::GROUPS];
  P::ints(wv, x, c, o);
#pragma unroll
  for (int g = 0; g < P::GROUPS; g++) out[P::group_index(c, g)] = o[g];
}

// ---- Q6_K x Q8_K ---------------------------------------------------------------------------------
// planes: ql[n][128] | qh[n][64] | scales[n][16] | d[n] (common.hpp).  A lane owns one 16-byte ql piece (8 lanes
// per super-block: one aligned 1 KiB request per wave): piece (h, a, p) = ql[64 h + 32 a + 16 p .. +16) holds the low
// 4 bits of two 16-element scale groups -- low nibbles: elements 128 h + 32 a + 16 p + [0, 16), high nibbles: the
// same + 64 -- and the matching 2-bit planes sit in qh[32 h + 16 p .. +16) at bit 2 a and 2 a + 4
// (buf_q6_k.rs:21-48).  6-bit values are rebuilt as bytes for v_dot4; the -32 offset is applied as -32 * bsum.
template <int R>
__global__ __launch_bounds__(256) void k_gemv_q6_k(const char* __restrict__ w, size_t off_qh, ActQ8_K act,
                                                   float* __restrict__ out, int m, int nsb) {
  const int lane = threadIdx.x & 63;
  const int wave = blockIdx.x * (blockDim.x >> 6) + wave_in_wg();
  const int row0 = wave * R;
  if (row0 >= m) return;
  float acc[R];
  rows_partial_q6k<R>(w, off_qh, act, row0, m, nsb, lane, acc);  // gemv_core.hpp
#pragma unroll
  for (int r = 0; r < R; r++) {
    float s = wave_sum_f32(acc[r]);
    if (lane == 0 && row0 + r < m) out[row0 + r] = s;
  }
}

// ---- Q8_K x Q8_K ---------------------------------------------------------------------------------
// planes: qs[n][256] | d[n] f32.  A lane owns one 32-element group; the 8 lanes of a super-block add
// their integer partials (exact) before the single f32 scaling `sum_i as f32 * d_a * d_b`.
template <int R>
__global__ __launch_bounds__(256) void k_gemv_q8_k(const i32x4* __restrict__ wq, const float* __restrict__ wd,
                                                   ActQ8_K act, float* __restrict__ out, int m, int nsb) {
  const int lane = threadIdx.x & 63;
  const int wave = blockIdx.x * (blockDim.x >> 6) + wave_in_wg();
  const int row0 = wave * R;
  if (row0 >= m) return;
  float acc[R];
#pragma unroll
  for (int r = 0; r < R; r++) acc[r] = 0.f;
  const int ngroups = nsb * 8;
  for (int g0 = 0; g0 < ngroups; g0 += 64) {
    const int g = g0 + lane;
    const bool live = g < ngroups;
    const int gg = live ? g : ngroups - 1;
    const int sb = gg >> 3;
    i32x4 x0 = act.q[2 * (size_t)gg], x1 = act.q[2 * (size_t)gg + 1];
    float d8 = act.d[sb];
#pragma unroll
    for (int r = 0; r < R; r++) {
      int row = row0 + r < m ? row0 + r : m - 1;
      size_t gi = (size_t)row * ngroups + gg;
      i32x4 q0 = __builtin_nontemporal_load(wq + 2 * gi);
      i32x4 q1 = __builtin_nontemporal_load(wq + 2 * gi + 1);
      int si = live ? dot_i8x32(q0, q1, x0, x1) : 0;
      si += __shfl_xor(si, 1, 64);
      si += __shfl_xor(si, 2, 64);
      si += __shfl_xor(si, 4, 64);
      if (live && (lane & 7) == 0) acc[r] += ((float)si * wd[(size_t)row * nsb + sb]) * d8;  // buf_q8_k.rs:220
    }
  }
#pragma unroll
  for (int r = 0; r < R; r++) {
    float s = wave_sum_f32(acc[r]);
    if (lane == 0 && row0 + r < m) out[row0 + r] = s;
  }
}

// ---- F32 / F16 weights ---------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_gemv_f32(const float* __restrict__ w, const float* __restrict__ x,
                                                  float* __restrict__ out, int m, int k) {
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * (blockDim.x >> 6) + wave_in_wg();
  if (row >= m) return;
  const float* wr = w + (size_t)row * k;
  float acc = 0.f;
  for (int i = lane; i < k; i += 64) acc += wr[i] * x[i];
  acc = wave_sum_f32(acc);
  if (lane == 0) out[row] = acc;
}
__global__ __launch_bounds__(256) void k_gemv_f16(const unsigned short* __restrict__ w,
                                                  const unsigned short* __restrict__ x16, float* __restrict__ out,
                                                  int m, int k) {
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * (blockDim.x >> 6) + wave_in_wg();
  if (row >= m) return;
  const unsigned short* wr = w + (size_t)row * k;
  float acc = 0.f;
  for (int i = lane; i < k; i += 64) acc += h2f(wr[i]) * h2f(x16[i]);
  acc = wave_sum_f32(acc);
  if (lane == 0) out[row] = acc;
}

// ---- host launcher -------------------------------------------------------------------------------
template <typename F>
static void launch_rows(hipStream_t st, int m, int n_cu, F&& f) {
  // R rows per wave: enough waves to cover the chip a few times over, otherwise fewer rows per wave
  // (lab: R=2 is best from ~14k rows, R=1 below).
  int R = m >= 8192 ? 2 : 1;
  int waves = (m + R - 1) / R;
  int tpb = 128;  // 2 waves per workgroup (lab: 64/128/256 within noise; 128 best on the classifier)
  int wpb = tpb / 64;
  int grid = (waves + wpb - 1) / wpb;
  f(R, grid, tpb);
  (void)n_cu;
}

int launch_gemv(crabml_hip_device* dev, const crabml_hip_buf* w, size_t m_, size_t k_, const void* act, size_t b,
                float* out, crabml_hip_device::ProfRec* rec0, bool fused_add) {
  hipStream_t st = dev->stream;
  const int m = (int)m_, k = (int)k_;
  const char* wp = (const char*)w->ptr;
  const uint32_t qt = vec_dot_rhs_dtype(w->dtype);
  const ActLayout al = act_layout(qt, k_);
  // F32 weights take the dense f32 rhs as is (row stride k*4); quantized planes are padded per row
  const size_t act_stride = qt == CRABML_HIP_F32 ? k_ * 4 : al.total;
  // a real batch (prefill): the weights are streamed once through the matrix cores instead of once per row
  if (b >= 16 && launch_gemm_mfma(dev, w, m_, k_, act, b, out, rec0, nullptr, fused_add)) return 0;
  for (size_t bi = 0; bi < b; bi++) {
    const char* ap = (const char*)act + bi * act_stride;
    float* o = out + bi * m_;
    crabml_hip_device::ProfRec* rec = bi == 0 ? rec0 : nullptr;
    switch (w->dtype) {
      case CRABML_HIP_Q4_0: {
        ActQ8_0 a{(const i32x4*)ap, (const unsigned short*)(ap + al.off_d), (const int*)(ap + al.off_aux)};
        const int nb = k / 32;
        launch_rows(st, m, dev->n_cu, [&](int R, int grid, int tpb) {
          if (R == 2)
            launch_k(st, rec, k_gemv_q4_0<2>, dim3(grid), dim3(tpb), 0, (const i32x4*)wp, (const unsigned short*)(wp + w->wl.off_scale), a, o, m, nb);
          else
            launch_k(st, rec, k_gemv_q4_0<1>, dim3(grid), dim3(tpb), 0, (const i32x4*)wp, (const unsigned short*)(wp + w->wl.off_scale), a, o, m, nb);
        });
        break;
      }
      case CRABML_HIP_Q8_0: {
        ActQ8_0 a{(const i32x4*)ap, (const unsigned short*)(ap + al.off_d), (const int*)(ap + al.off_aux)};
        const int nb = k / 32;
        launch_rows(st, m, dev->n_cu, [&](int R, int grid, int tpb) {
          if (R == 2)
            launch_k(st, rec, k_gemv_q8_0<2>, dim3(grid), dim3(tpb), 0, (const i32x4*)wp, (const unsigned short*)(wp + w->wl.off_scale), a, o, m, nb);
          else
            launch_k(st, rec, k_gemv_q8_0<1>, dim3(grid), dim3(tpb), 0, (const i32x4*)wp, (const unsigned short*)(wp + w->wl.off_scale), a, o, m, nb);
        });
        break;
      }
      case CRABML_HIP_Q4_1: {
        ActQ8_1 a{(const i32x4*)ap, (const unsigned short*)(ap + al.off_d), (const unsigned short*)(ap + al.off_aux)};
        const int nb = k / 32;
        launch_rows(st, m, dev->n_cu, [&](int R, int grid, int tpb) {
          if (R == 2)
            launch_k(st, rec, k_gemv_q4_1<2>, dim3(grid), dim3(tpb), 0, (const i32x4*)wp, (const unsigned*)(wp + w->wl.off_scale), a, o, m, nb);
          else
            launch_k(st, rec, k_gemv_q4_1<1>, dim3(grid), dim3(tpb), 0, (const i32x4*)wp, (const unsigned*)(wp + w->wl.off_scale), a, o, m, nb);
        });
        break;
      }
      case CRABML_HIP_Q4_K: {
        const ActQ8_K a = act_q8k_at(ap, al.off_d, al.off_aux, al.off_p);
        const int nsb = k / 256;
        launch_rows(st, m, dev->n_cu, [&](int R, int grid, int tpb) {
          if (R == 2)
            launch_k(st, rec, k_gemv_q4_k<2>, dim3(grid), dim3(tpb), 0, (const i32x4*)wp, (const i32x4*)(wp + w->wl.off_scale), a, o, m, nsb);
          else
            launch_k(st, rec, k_gemv_q4_k<1>, dim3(grid), dim3(tpb), 0, (const i32x4*)wp, (const i32x4*)(wp + w->wl.off_scale), a, o, m, nsb);
        });
        break;
      }
      case CRABML_HIP_Q5_K: {
        const ActQ8_K a = act_q8k_at(ap, al.off_d, al.off_aux, al.off_p);
        const int nsb = k / 256;
        launch_rows(st, m, dev->n_cu, [&](int R, int grid, int tpb) {
          if (R == 2)
            launch_k(st, rec, k_gemv_q5_k<2>, dim3(grid), dim3(tpb), 0, wp, w->wl.off_scale, a, o, m, nsb);
          else
            launch_k(st, rec, k_gemv_q5_k<1>, dim3(grid), dim3(tpb), 0, wp, w->wl.off_scale, a, o, m, nsb);
        });
        break;
      }
      case CRABML_HIP_Q6_K: {
        const ActQ8_K a = act_q8k_at(ap, al.off_d, al.off_aux, al.off_p);
        const int nsb = k / 256;
        launch_rows(st, m, dev->n_cu, [&](int R, int grid, int tpb) {
          if (R == 2)
            launch_k(st, rec, k_gemv_q6_k<2>, dim3(grid), dim3(tpb), 0, wp, w->wl.off_scale, a, o, m, nsb);
          else
            launch_k(st, rec, k_gemv_q6_k<1>, dim3(grid), dim3(tpb), 0, wp, w->wl.off_scale, a, o, m, nsb);
        });
        break;
      }
      case CRABML_HIP_Q8_K: {
        const ActQ8_K a = act_q8k_at(ap, al.off_d, al.off_aux, al.off_p);
        const int nsb = k / 256;
        launch_rows(st, m, dev->n_cu, [&](int R, int grid, int tpb) {
          if (R == 2)
            launch_k(st, rec, k_gemv_q8_k<2>, dim3(grid), dim3(tpb), 0, (const i32x4*)wp, (const float*)(wp + w->wl.off_scale), a, o, m, nsb);
          else
            launch_k(st, rec, k_gemv_q8_k<1>, dim3(grid), dim3(tpb), 0, (const i32x4*)wp, (const float*)(wp + w->wl.off_scale), a, o, m, nsb);
        });
        break;
      }
      case CRABML_HIP_Q5_0: {
        ActQ8_0 a{(const i32x4*)ap, (const unsigned short*)(ap + al.off_d), (const int*)(ap + al.off_aux)};
        launch_rows(st, m, dev->n_cu, [&](int R, int grid, int tpb) {
          if (R == 2)
            launch_k(st, rec, k_gemv_pieces<PieceQ5_0, 2>, dim3(grid), dim3(tpb), 0, wp, w->wl.off_scale, w->wl.n_blocks, a, o, m, k / 32);
          else
            launch_k(st, rec, k_gemv_pieces<PieceQ5_0, 1>, dim3(grid), dim3(tpb), 0, wp, w->wl.off_scale, w->wl.n_blocks, a, o, m, k / 32);
        });
        break;
      }
      case CRABML_HIP_Q5_1: {
        ActQ8_1 a{(const i32x4*)ap, (const unsigned short*)(ap + al.off_d), (const unsigned short*)(ap + al.off_aux)};
        launch_rows(st, m, dev->n_cu, [&](int R, int grid, int tpb) {
          if (R == 2)
            launch_k(st, rec, k_gemv_pieces<PieceQ5_1, 2>, dim3(grid), dim3(tpb), 0, wp, w->wl.off_scale, w->wl.n_blocks, a, o, m, k / 32);
          else
            launch_k(st, rec, k_gemv_pieces<PieceQ5_1, 1>, dim3(grid), dim3(tpb), 0, wp, w->wl.off_scale, w->wl.n_blocks, a, o, m, k / 32);
        });
        break;
      }
      case CRABML_HIP_Q2_K: {
        const ActQ8_K a = act_q8k_at(ap, al.off_d, al.off_aux, al.off_p);
        launch_rows(st, m, dev->n_cu, [&](int R, int grid, int tpb) {
          if (R == 2)
            launch_k(st, rec, k_gemv_pieces<PieceQ2_K, 2>, dim3(grid), dim3(tpb), 0, wp, w->wl.off_scale, w->wl.n_blocks, a, o, m, k / 256);
          else
            launch_k(st, rec, k_gemv_pieces<PieceQ2_K, 1>, dim3(grid), dim3(tpb), 0, wp, w->wl.off_scale, w->wl.n_blocks, a, o, m, k / 256);
        });
        break;
      }
      case CRABML_HIP_Q3_K: {
        const ActQ8_K a = act_q8k_at(ap, al.off_d, al.off_aux, al.off_p);
        launch_rows(st, m, dev->n_cu, [&](int R, int grid, int tpb) {
          if (R == 2)
            launch_k(st, rec, k_gemv_pieces<PieceQ3_K, 2>, dim3(grid), dim3(tpb), 0, wp, w->wl.off_scale, w->wl.n_blocks, a, o, m, k / 256);
          else
            launch_k(st, rec, k_gemv_pieces<PieceQ3_K, 1>, dim3(grid), dim3(tpb), 0, wp, w->wl.off_scale, w->wl.n_blocks, a, o, m, k / 256);
        });
        break;
      }
      case CRABML_HIP_F32: {
        int grid = (m + 3) / 4;
        launch_k(st, rec, k_gemv_f32, dim3(grid), dim3(256), 0, (const float*)wp, (const float*)ap, o, m, k);
        break;
      }
      case CRABML_HIP_F16: {
        int grid = (m + 3) / 4;
        launch_k(st, rec, k_gemv_f16, dim3(grid), dim3(256), 0, (const unsigned short*)wp, (const unsigned short*)ap, o, m, k);
        break;
      }
      default:
        return set_error(dev, CRABML_HIP_TENSOR_ERROR, "matmul_vec: unsupported weight dtype %u", w->dtype);
    }
  }
  return 0;
}

// ---- parity hook: the exact integer part per 32-element group, through the SAME unpack code ----------
__global__ void k_block_dots_32(const i32x4* __restrict__ wq, int wtype, ActQ8_0 a0, ActQ8_1 a1, size_t row_block0,
                                int nb, int* __restrict__ out) {
  int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= nb) return;
  size_t idx = row_block0 + b;
  if (wtype == CRABML_HIP_Q4_0) {
    out[b] = dot_q4_0(wq[idx], a0.q[2 * b], a0.q[2 * b + 1], a0.isum[b]);
  } else if (wtype == CRABML_HIP_Q8_0) {
    out[b] = dot_i8x32(wq[2 * idx], wq[2 * idx + 1], a0.q[2 * b], a0.q[2 * b + 1]);
  } else {
    out[b] = dot_u4(wq[idx], a1.q[2 * b], a1.q[2 * b + 1]);
  }
}
__global__ void k_block_dots_k(const unsigned char* __restrict__ w, int wtype, ActQ8_K a, size_t row_sb0, int nsb,
                               int* __restrict__ out, size_t off_qh) {
  int g = blockIdx.x * blockDim.x + threadIdx.x;  // 32-element group
  if (g >= nsb * 8) return;
  int sb = g >> 3;
  if (wtype == CRABML_HIP_Q4_K || wtype == CRABML_HIP_Q5_K) {
    int p = (g & 7) >> 1, hi_half = g & 1;
    const unsigned char* qs = w + (row_sb0 + sb) * 128;
    i32x4 qa = *(const i32x4*)(qs + p * 32), qb = *(const i32x4*)(qs + 16 + p * 32);
    i32x4 ha = {0, 0, 0, 0}, hb = {0, 0, 0, 0};  // Q5_K: the fifth bit of this group's 32 levels, moved to bit 4 of their bytes
    if (wtype == CRABML_HIP_Q5_K) {
      const unsigned char* qh = w + off_qh + (row_sb0 + sb) * 32;
      ha = *(const i32x4*)qh;
      hb = *(const i32x4*)(qh + 16);
    }
    // (Q4_K: weights and activations class-major inside the 32-group, common.hpp -- the group's sum is the same)
    const i32x4* xq = (wtype == CRABML_HIP_Q4_K ? a.qp : a.q) + (size_t)sb * 16 + p * 4 + hi_half * 2;
    i32x4 x0 = xq[0], x1 = xq[1];
    int s = 0;
    for (int i = 0; i < 4; i++) {
      int wa = hi_half ? (qa[i] >> 4) : qa[i], wb = hi_half ? (qb[i] >> 4) : qb[i];
      const int fa = (int)((((unsigned)ha[i] >> (2 * p + hi_half)) & 0x01010101u) << 4);
      const int fb = (int)((((unsigned)hb[i] >> (2 * p + hi_half)) & 0x01010101u) << 4);
      s = __builtin_amdgcn_sdot4((wa & 0x0F0F0F0F) | fa, x0[i], s, false);
      s = __builtin_amdgcn_sdot4((wb & 0x0F0F0F0F) | fb, x1[i], s, false);
    }
    out[g] = s;
  } else {
    const i32x4* wq = (const i32x4*)w;
    size_t gi = row_sb0 * 8 + g;
    out[g] = dot_i8x32(wq[2 * gi], wq[2 * gi + 1], a.q[2 * (size_t)g], a.q[2 * (size_t)g + 1]);
  }
}

// Q6_K: the exact integer part per 16-element scale group, sum (q6 - 32) * q8, through the production unpack
__global__ void k_block_dots_q6k(const char* __restrict__ w, size_t off_qh, ActQ8_K act, size_t row_sb0, int nsb,
                                 int* __restrict__ out) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;  // ql piece
  if (c >= nsb * 8) return;
  const int sb = c >> 3, h = (c >> 2) & 1, a = (c >> 1) & 1, p = c & 1;
  const int gi = 8 * h + p + 2 * a;
  const size_t blk = row_sb0 + sb;
  const i32x4 qv = ((const i32x4*)w)[blk * 8 + (c & 7)];
  const i32x4 hv = ((const i32x4*)(w + off_qh))[blk * 4 + 2 * h + p];
  const i32x4* xq = act.q + (size_t)sb * 16 + gi;
  const i32x4 xl = xq[0], xh = xq[4];
  int lo = 0, hi = 0;
  for (int i = 0; i < 4; i++) {
    const unsigned q = (unsigned)qv[i], hb = (unsigned)hv[i] >> (2 * a);
    lo = __builtin_amdgcn_sdot4((int)((q & 0x0F0F0F0Fu) | ((hb & 0x03030303u) << 4)), xl[i], lo, false);
    hi = __builtin_amdgcn_sdot4((int)(((q >> 4) & 0x0F0F0F0Fu) | (((hb >> 4) & 0x03030303u) << 4)), xh[i], hi, false);
  }
  out[sb * 16 + gi] = lo - 32 * (int)act.bsums[sb * 16 + gi];
  out[sb * 16 + gi + 4] = hi - 32 * (int)act.bsums[sb * 16 + gi + 4];
}

// ---- parity hook: the integers of the PRODUCTION K-quant loops (rows_partial_q4k / rows_partial_q6k, DBG instantiation) --
// One wave walks row `row` exactly as k_gemv_q4_k<1> / k_gemv_q6_k<1> do; out[2 c], out[2 c + 1] = the two integers piece c
// hands to its float part (Q4_K: isum, msum; Q6_K: scale_lo * dot_lo, scale_hi * dot_hi, the -32 offset included).
__global__ __launch_bounds__(64) void k_piece_ints_q4k(const i32x4* __restrict__ wq, const i32x4* __restrict__ wh, ActQ8_K act, int row,
                                                       int m, int nsb, int* __restrict__ out, float* __restrict__ fout) {
  float acc[1];
  rows_partial_q4k<1, true, true>(wq, wh, act, row, m, nsb, (int)threadIdx.x, acc, 0, out);
  const float s = wave_sum_f32(acc[0]);
  if (threadIdx.x == 0) *fout = s;
}
__global__ __launch_bounds__(64) void k_piece_ints_q4k_lds(const i32x4* __restrict__ wq, const i32x4* __restrict__ wh, ActQ8_K act, int row,
                                                           int m, int nsb, int* __restrict__ out, float* __restrict__ fout) {
  float acc[1];  // the whole-header form the LDS-staged kernels use (HDR_DPP = false)
  rows_partial_q4k<1, false, true>(wq, wh, act, row, m, nsb, (int)threadIdx.x, acc, 0, out);
  const float s = wave_sum_f32(acc[0]);
  if (threadIdx.x == 0) *fout = s;
}
__global__ __launch_bounds__(64) void k_piece_ints_q6k(const char* __restrict__ w, size_t off_qh, ActQ8_K act, int row, int m, int nsb,
                                                       int* __restrict__ out, float* __restrict__ fout) {
  float acc[1];
  rows_partial_q6k<1, true>(w, off_qh, act, row, m, nsb, (int)threadIdx.x, acc, out);
  const float s = wave_sum_f32(acc[0]);
  if (threadIdx.x == 0) *fout = s;
}
int launch_piece_ints(crabml_hip_device* dev, const crabml_hip_buf* w, size_t m, size_t k, size_t row, const void* act, int variant,
                      int32_t* out, float* fout) {
  const char* wp = (const char*)w->ptr;
  const char* ap = (const char*)act;
  const ActLayout al = act_layout(CRABML_HIP_Q8_K, k);
  const ActQ8_K a = act_q8k_at(ap, al.off_d, al.off_aux, al.off_p);
  const int nsb = (int)(k / 256);
  if (w->dtype == CRABML_HIP_Q4_K) {
    if (variant == 0)
      k_piece_ints_q4k<<<1, 64, 0, dev->stream>>>((const i32x4*)wp, (const i32x4*)(wp + w->wl.off_scale), a, (int)row, (int)m, nsb, out, fout);
    else
      k_piece_ints_q4k_lds<<<1, 64, 0, dev->stream>>>((const i32x4*)wp, (const i32x4*)(wp + w->wl.off_scale), a, (int)row, (int)m, nsb, out, fout);
  } else if (w->dtype == CRABML_HIP_Q6_K) {
    k_piece_ints_q6k<<<1, 64, 0, dev->stream>>>(wp, w->wl.off_scale, a, (int)row, (int)m, nsb, out, fout);
  } else {
    return set_error(dev, CRABML_HIP_TENSOR_ERROR, "debug_superblock_ints: Q4_K / Q6_K weights only");
  }
  return 0;
}

void launch_block_dots(hipStream_t st, const crabml_hip_buf* w, size_t k, size_t row, const void* act, int32_t* out) {
  const char* wp = (const char*)w->ptr;
  const char* ap = (const char*)act;
  const uint32_t qt = vec_dot_rhs_dtype(w->dtype);
  const ActLayout al = act_layout(qt, k);
  if (w->dtype == CRABML_HIP_Q5_0) {
    ActQ8_0 a{(const i32x4*)ap, (const unsigned short*)(ap + al.off_d), (const int*)(ap + al.off_aux)};
    const int nb = (int)(k / 32);
    k_block_dots_pieces<PieceQ5_0><<<(nb + 63) / 64, 64, 0, st>>>(wp, w->wl.off_scale, w->wl.n_blocks, a, row * nb, nb, out);
  } else if (w->dtype == CRABML_HIP_Q5_1) {
    ActQ8_1 a{(const i32x4*)ap, (const unsigned short*)(ap + al.off_d), (const unsigned short*)(ap + al.off_aux)};
    const int nb = (int)(k / 32);
    k_block_dots_pieces<PieceQ5_1><<<(nb + 63) / 64, 64, 0, st>>>(wp, w->wl.off_scale, w->wl.n_blocks, a, row * nb, nb, out);
  } else if (w->dtype == CRABML_HIP_Q2_K || w->dtype == CRABML_HIP_Q3_K) {
    const ActQ8_K a = act_q8k_at(ap, al.off_d, al.off_aux, al.off_p);
    const int nsb = (int)(k / 256);
    if (w->dtype == CRABML_HIP_Q2_K)
      k_block_dots_pieces<PieceQ2_K><<<(nsb * 4 + 63) / 64, 64, 0, st>>>(wp, w->wl.off_scale, w->wl.n_blocks, a, row * nsb, nsb * 4, out);
    else
      k_block_dots_pieces<PieceQ3_K><<<(nsb * 4 + 63) / 64, 64, 0, st>>>(wp, w->wl.off_scale, w->wl.n_blocks, a, row * nsb, nsb * 4, out);
  } else if (w->dtype == CRABML_HIP_Q6_K) {
    const ActQ8_K a = act_q8k_at(ap, al.off_d, al.off_aux, al.off_p);
    int nsb = (int)(k / 256);
    k_block_dots_q6k<<<(nsb * 8 + 63) / 64, 64, 0, st>>>(wp, w->wl.off_scale, a, row * nsb, nsb, out);
  } else if (w->dtype == CRABML_HIP_Q4_K || w->dtype == CRABML_HIP_Q5_K || w->dtype == CRABML_HIP_Q8_K) {
    const ActQ8_K a = act_q8k_at(ap, al.off_d, al.off_aux, al.off_p);
    int nsb = (int)(k / 256);
    k_block_dots_k<<<(nsb * 8 + 63) / 64, 64, 0, st>>>((const unsigned char*)wp, (int)w->dtype, a, row * nsb, nsb, out, w->wl.off_scale);
  } else {
    ActQ8_0 a0{(const i32x4*)ap, (const unsigned short*)(ap + al.off_d), (const int*)(ap + al.off_aux)};
    ActQ8_1 a1{(const i32x4*)ap, (const unsigned short*)(ap + al.off_d), (const unsigned short*)(ap + al.off_aux)};
    int nb = (int)(k / 32);
    k_block_dots_32<<<(nb + 63) / 64, 64, 0, st>>>((const i32x4*)wp, (int)w->dtype, a0, a1, row * nb, nb, out);
  }
}

}  // namespace crabml_hip

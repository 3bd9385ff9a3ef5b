// gemv_strict.hip -- strict-order matmul_vec (CRABML_HIP_FLAG_STRICT_ORDER): a parity instrument.
//
// One thread per output row walks the row's blocks in order and evaluates exactly the reference's
// scalar loops (vec_dot_*_fallback: buf_q4_0.rs:240-253, buf_q8_0.rs:275-286, buf_q4_1.rs:266-280,
// buf_q4_k.rs:192-277, buf_q8_k.rs:211-224, buf_f32.rs:19-27, buf_f16.rs:83-97) -- same integer sums,
// same f32 expression, same association -- on the same device planes the fast kernels read.  Logits are
// then bit-identical to the default (non-SIMD) build of the reference, which turns the end-to-end parity
// check into an equality test.  Not a performance path (uncoalesced by construction).
#include <cstdlib>

#include "devutil.hpp"
#include "gemv_core.hpp"
#include "kernels.hpp"

namespace crabml_hip {

__global__ __launch_bounds__(64) void k_gemv_strict(const char* __restrict__ w, int dtype, size_t off_scale,
                                                    const char* __restrict__ act, size_t off_d, size_t off_aux,
                                                    float* __restrict__ out, const float* add, int m, int k) {
  const int row = blockIdx.x * blockDim.x + threadIdx.x;
  if (row >= m) return;
  float sumf = 0.0f;
  switch (dtype) {
    case CRABML_HIP_Q4_0: {
      const int nb = k / 32;
      const unsigned short* wd = (const unsigned short*)(w + off_scale);
      const unsigned short* xd = (const unsigned short*)(act + off_d);
      for (int b = 0; b < nb; b++) {
        const unsigned char* qs = (const unsigned char*)w + ((size_t)row * nb + b) * 16;
        const signed char* xq = (const signed char*)act + (size_t)b * 32;
        int sumi = 0;
        for (int j = 0; j < 16; j++) {
          int v0 = (int)(qs[j] & 0x0F) - 8, v1 = (int)(qs[j] >> 4) - 8;
          sumi += v0 * (int)xq[j] + v1 * (int)xq[j + 16];
        }
        sumf += (float)sumi * h2f(wd[(size_t)row * nb + b]) * h2f(xd[b]);
      }
      break;
    }
    case CRABML_HIP_Q8_0: {
      const int nb = k / 32;
      const unsigned short* wd = (const unsigned short*)(w + off_scale);
      const unsigned short* xd = (const unsigned short*)(act + off_d);
      for (int b = 0; b < nb; b++) {
        const signed char* qs = (const signed char*)w + ((size_t)row * nb + b) * 32;
        const signed char* xq = (const signed char*)act + (size_t)b * 32;
        int sumi = 0;
        for (int j = 0; j < 32; j++) sumi += (int)qs[j] * (int)xq[j];
        sumf += (float)sumi * h2f(wd[(size_t)row * nb + b]) * h2f(xd[b]);
      }
      break;
    }
    case CRABML_HIP_Q4_1: {
      const int nb = k / 32;
      const unsigned* wdm = (const unsigned*)(w + off_scale);
      const unsigned short* xd = (const unsigned short*)(act + off_d);
      const unsigned short* xs = (const unsigned short*)(act + off_aux);
      for (int b = 0; b < nb; b++) {
        const unsigned char* qs = (const unsigned char*)w + ((size_t)row * nb + b) * 16;
        const signed char* xq = (const signed char*)act + (size_t)b * 32;
        int sumi = 0;
        for (int j = 0; j < 16; j++) {
          int v0 = (int)(qs[j] & 0x0F), v1 = (int)((qs[j] >> 4) & 0x0F);
          sumi += v0 * (int)xq[j] + v1 * (int)xq[j + 16];
        }
        unsigned dm = wdm[(size_t)row * nb + b];
        unsigned short dw = (unsigned short)(dm & 0xffffu), mw = (unsigned short)(dm >> 16);
        sumf += h2f(h_mul(dw, xd[b])) * (float)sumi + h2f(h_mul(mw, xs[b]));
      }
      break;
    }
    case CRABML_HIP_Q5_0: {  // buf_q5_0.rs:143-161; planes qs | qh | d with n = off_scale / 16 blocks
      const int nb = k / 32;
      const size_t n = off_scale / 16;
      const unsigned* wqh = (const unsigned*)(w + off_scale);
      const unsigned short* wd = (const unsigned short*)(w + off_scale + n * 4);
      const unsigned short* xd = (const unsigned short*)(act + off_d);
      for (int b = 0; b < nb; b++) {
        const size_t blk = (size_t)row * nb + b;
        const unsigned char* qs = (const unsigned char*)w + blk * 16;
        const signed char* xq = (const signed char*)act + (size_t)b * 32;
        const unsigned qh = wqh[blk];
        int sumi = 0;
        for (int j = 0; j < 16; j++) {
          const int x0 = (int)((qs[j] & 0x0F) | (((qh >> j) & 1u) << 4)) - 16, x1 = (int)((qs[j] >> 4) | (((qh >> (j + 16)) & 1u) << 4)) - 16;
          sumi += x0 * (int)xq[j] + x1 * (int)xq[j + 16];
        }
        sumf += (float)sumi * h2f(wd[blk]) * h2f(xd[b]);
      }
      break;
    }
    case CRABML_HIP_Q5_1: {  // buf_q5_1.rs:141-160; planes qs | (d, m, qh)
      const int nb = k / 32;
      const unsigned* rec = (const unsigned*)(w + off_scale);
      const unsigned short* xd = (const unsigned short*)(act + off_d);
      const unsigned short* xs = (const unsigned short*)(act + off_aux);
      for (int b = 0; b < nb; b++) {
        const size_t blk = (size_t)row * nb + b;
        const unsigned char* qs = (const unsigned char*)w + blk * 16;
        const signed char* xq = (const signed char*)act + (size_t)b * 32;
        const unsigned dm = rec[2 * blk], qh = rec[2 * blk + 1];
        int sumi = 0;
        for (int j = 0; j < 16; j++) {
          const int x0 = (int)((qs[j] & 0x0F) | (((qh >> j) & 1u) << 4)), x1 = (int)((qs[j] >> 4) | (((qh >> (j + 16)) & 1u) << 4));
          sumi += x0 * (int)xq[j] + x1 * (int)xq[j + 16];
        }
        sumf += (float)sumi * h2f(h_mul((unsigned short)(dm & 0xffffu), xd[b])) + h2f(h_mul((unsigned short)(dm >> 16), xs[b]));
      }
      break;
    }
    case CRABML_HIP_Q2_K: {  // buf_q2_k.rs:216-258; planes qs | scales | (d, dmin) with n = off_scale / 64 blocks
      const int nsb = k / 256;
      const size_t n = off_scale / 64;
      const float* xd = (const float*)(act + off_d);
      const short* bsums = (const short*)(act + off_aux);
      for (int sb = 0; sb < nsb; sb++) {
        const size_t blk = (size_t)row * nsb + sb;
        const unsigned char* q2 = (const unsigned char*)w + blk * 64;
        const unsigned char* sc = (const unsigned char*)w + off_scale + blk * 16;
        const unsigned dm = ((const unsigned*)(w + off_scale + n * 16))[blk];
        const signed char* q8 = (const signed char*)act + (size_t)sb * 256;
        int summs = 0;  // an i16 in the reference: exact here (the oracle counts the inputs on which the two differ)
        for (int j = 0; j < 16; j++) summs += (int)bsums[sb * 16 + j] * (int)(sc[j] >> 4);
        const float dall = xd[sb] * h2f((unsigned short)(dm & 0xffffu)), dmin = xd[sb] * h2f((unsigned short)(dm >> 16));
        int isum = 0, is = 0;
        for (int half = 0; half < 2; half++)
          for (int shift = 0; shift < 8; shift += 2)
            for (int h = 0; h < 2; h++) {
              const int d = sc[is] & 0xF;
              int isuml = 0;
              for (int l = 0; l < 16; l++) isuml += (int)q8[16 * is + l] * (int)((q2[32 * half + 16 * h + l] >> shift) & 3);
              isum += d * isuml;
              is++;
            }
        sumf += dall * (float)isum - dmin * (float)summs;
      }
      break;
    }
    case CRABML_HIP_Q3_K: {  // buf_q3_k.rs:238-329: eight i32 lanes per block (element e feeds lane e % 8), eight f32 sums
      const int nsb = k / 256;
      const size_t n = off_scale / 64;
      const float* xd = (const float*)(act + off_d);
      float sums[8];
      for (int l = 0; l < 8; l++) sums[l] = 0.0f;
      for (int sb = 0; sb < nsb; sb++) {
        const size_t blk = (size_t)row * nsb + sb;
        const unsigned char* q3 = (const unsigned char*)w + blk * 64;
        const unsigned char* hm = (const unsigned char*)w + off_scale + blk * 32;
        const unsigned char* sd = (const unsigned char*)w + off_scale + n * 32 + blk * 16;
        const signed char* q8 = (const signed char*)act + (size_t)sb * 256;
        int aux32[8];
        for (int l = 0; l < 8; l++) aux32[l] = 0;
        for (int g = 0; g < 16; g++) {
          const int half = g >> 3, s = (g & 7) >> 1, h = g & 1;
          const int lo = g < 8 ? (sd[g] & 0xF) : (sd[g - 8] >> 4), hi = (sd[8 + (g & 3)] >> (2 * (g >> 2))) & 3;
          const int scale = (lo | (hi << 4)) - 32;
          for (int e = 0; e < 16; e++) {
            const int pos = 16 * h + e;  // position within the half's 32 qs bytes / the 32 hmask bytes
            const int a8 = (int)((q3[32 * half + pos] >> (2 * s)) & 3) - (((hm[pos] >> (4 * half + s)) & 1) ? 0 : 4);
            aux32[e & 7] += scale * ((int)q8[16 * g + e] * a8);
          }
        }
        unsigned short dh;
        __builtin_memcpy(&dh, sd + 12, 2);
        const float d = h2f(dh) * xd[sb];
        for (int l = 0; l < 8; l++) sums[l] += d * (float)aux32[l];
      }
      sumf = sums[0];
      for (int l = 1; l < 8; l++) sumf = sumf + sums[l];
      break;
    }
    case CRABML_HIP_Q4_K:
    case CRABML_HIP_Q5_K: {  // buf_q5_k.rs:229-325 = buf_q4_k.rs:192-277 + the fifth bit (planes qs | qh | hdr, n = off_scale / 128 blocks)
      const bool q5 = dtype == CRABML_HIP_Q5_K;
      const size_t n5 = off_scale / 128;
      const int nsb = k / 256;
      const float* xd = (const float*)(act + off_d);
      const short* bsums = (const short*)(act + off_aux);
      float sums[8];
      for (int l = 0; l < 8; l++) sums[l] = 0.0f;
      for (int sb = 0; sb < nsb; sb++) {
        const unsigned char* hdr = (const unsigned char*)w + off_scale + (q5 ? n5 * 32 : 0) + ((size_t)row * nsb + sb) * 16;
        const unsigned char* qh = (const unsigned char*)w + off_scale + ((size_t)row * nsb + sb) * 32;  // (Q5_K only)
        const signed char* q8 = (const signed char*)act + (size_t)sb * 256;
        unsigned short dh, mh;
        __builtin_memcpy(&dh, hdr, 2);
        __builtin_memcpy(&mh, hdr + 2, 2);
        const unsigned char* q4 = (const unsigned char*)w + ((size_t)row * nsb + sb) * 128;
        float aux32[8];
        for (int l = 0; l < 8; l++) aux32[l] = 0.0f;
        int scales[8], mins[8];
        {
          unsigned u0, u1, u2;  // the re-packed (scale, min) fields (common.hpp)
          __builtin_memcpy(&u0, hdr + 4, 4);
          __builtin_memcpy(&u1, hdr + 8, 4);
          __builtin_memcpy(&u2, hdr + 12, 4);
          for (int pp = 0; pp < 4; pp++) {
            const unsigned f = q4k_pair_field(u0, u1, u2, pp);
            scales[2 * pp] = (int)(f & 63u);
            scales[2 * pp + 1] = (int)((f >> 6) & 63u);
            mins[2 * pp] = (int)((f >> 12) & 63u);
            mins[2 * pp + 1] = (int)(f >> 18);
          }
        }
        int sumi = 0;
        for (int j = 0; j < 16; j++) sumi += (int)bsums[sb * 16 + j] * mins[j / 2];
        for (int is = 0; is < 8; is++) {
          const float scale = (float)scales[is];
          const int c = is >> 1, hi = is & 1;
          for (int g = 0; g < 4; g++)
            for (int l = 0; l < 8; l++) {
              int e = 8 * g + l;  // element within the 32-wide sub-block
              unsigned char qb = q4[32 * c + (q5 ? e : q4k_perm_index(e))];  // (Q4_K: class-major qs plane, common.hpp)
              int a = hi ? (qb >> 4) : (qb & 0xF);
              if (q5 && ((qh[e] >> (2 * c + hi)) & 1)) a += 16;
              int prod = (int)q8[32 * is + e] * a;  // aux16
              aux32[l] += scale * (float)prod;
            }
        }
        const float d = h2f(dh) * xd[sb];
        for (int l = 0; l < 8; l++) sums[l] += d * aux32[l];
        const float dmin = h2f(mh) * xd[sb];
        sumf -= dmin * (float)sumi;
      }
      for (int l = 0; l < 8; l++) sumf += sums[l];
      break;
    }
    case CRABML_HIP_Q6_K: {  // buf_q6_k.rs:183-234: eight f32 lanes, element e feeds lane e % 8, blocks in order
      const int nsb = k / 256;
      const size_t n = off_scale / 128;
      const float* xd = (const float*)(act + off_d);
      float sums[8];
      for (int l = 0; l < 8; l++) sums[l] = 0.0f;
      for (int sb = 0; sb < nsb; sb++) {
        const size_t blk = (size_t)row * nsb + sb;
        const unsigned char* ql = (const unsigned char*)w + blk * 128;
        const unsigned char* qh = (const unsigned char*)w + off_scale + blk * 64;
        const signed char* sc = (const signed char*)w + off_scale + n * 64 + blk * 16;
        const signed char* q8 = (const signed char*)act + (size_t)sb * 256;
        float aux32[8];
        for (int l = 0; l < 8; l++) aux32[l] = 0.0f;
        for (int j = 0; j < 16; j++) {
          const float scale = (float)sc[j];
          for (int half8 = 0; half8 < 2; half8++)
            for (int l = 0; l < 8; l++) {
              const int e = 16 * j + 8 * half8 + l;  // element within the super-block
              const int idx = e / 128, r = e % 128, lq = r % 32, quarter = r / 32;
              const unsigned char lo = (quarter & 1) ? ql[64 * idx + lq + 32] : ql[64 * idx + lq];
              const int nib = quarter >= 2 ? (lo >> 4) : (lo & 0xF);
              const int hi2 = (qh[32 * idx + lq] >> (2 * quarter)) & 3;
              const int a8 = (nib | (hi2 << 4)) - 32;
              const int prod = (int)q8[e] * a8;  // aux16
              aux32[l] += scale * (float)prod;
            }
        }
        const float d = h2f(((const unsigned short*)(w + off_scale + n * 80))[blk]) * xd[sb];
        for (int l = 0; l < 8; l++) sums[l] += aux32[l] * d;
      }
      for (int l = 0; l < 8; l++) sumf += sums[l];
      break;
    }
    case CRABML_HIP_Q8_K: {
      const int nsb = k / 256;
      const float* wd = (const float*)(w + off_scale);
      const float* xd = (const float*)(act + off_d);
      for (int sb = 0; sb < nsb; sb++) {
        const signed char* qs = (const signed char*)w + ((size_t)row * nsb + sb) * 256;
        const signed char* xq = (const signed char*)act + (size_t)sb * 256;
        int s = 0;
        for (int j = 0; j < 256; j++) s += (int)qs[j] * (int)xq[j];
        sumf += (float)s * wd[(size_t)row * nsb + sb] * xd[sb];
      }
      break;
    }
    case CRABML_HIP_F32: {
      const float* wr = (const float*)w + (size_t)row * k;
      const float* x = (const float*)act;
      for (int i = 0; i < k; i++) sumf += wr[i] * x[i];
      break;
    }
    case CRABML_HIP_F16: {
      const unsigned short* wr = (const unsigned short*)w + (size_t)row * k;
      const unsigned short* x = (const unsigned short*)act;
      for (int i = 0; i < k; i++) sumf += h2f(wr[i]) * h2f(x[i]);
      break;
    }
    default: break;
  }
  out[row] = add ? sumf + add[row] : sumf;  // (add: the residual, x = matmul_out + x)
}

// ---- the same sums at streaming speed (round 4) -----------------------------------------------------------------------------------
// For the formats whose reference dot is ONE f32 term per block added in block order (Q4_0, Q8_0, Q4_1, Q5_0, Q5_1; per
// super-block: Q2_K, Q8_K) the order only matters for the final chain.  The terms are evaluated the way the fast kernels do it --
// coalesced 16-byte loads, one unit per lane, exact integers, the reference's f32 expression per block -- but instead of being
// added per lane and reduced through a tree they are parked in LDS, and one lane per row then adds them strictly in block order:
// sumf = 0; sumf += t_0; sumf += t_1; ...  Every output equals k_gemv_strict's bit for bit (tests/test_hip_gemv.py compares the two
// and the oracle).  The chain (nb dependent adds) runs in one lane while the other waves of the CU keep streaming.
// Q3_K .. Q6_K (eight lanes inside the super-block) have their own kernels below; dense F32 / F16 rows stay on k_gemv_strict.
template <int R>
__device__ __forceinline__ void exact_chain_store(const float* __restrict__ T, int nterms, int stride, int row0, int m, int lane,
                                                  float* __restrict__ out, const float* add) {
  __builtin_amdgcn_s_waitcnt(0xC07F);  // lgkmcnt(0): this wave's own LDS stores have landed (no other wave touches its region)
  __builtin_amdgcn_wave_barrier();
  if (lane < R && row0 + lane < m) {
    const float sumf = ordered_sum(T + (size_t)lane * stride, nterms);  // (rows of the term table are padded to 16 bytes)
    out[row0 + lane] = add ? sumf + add[row0 + lane] : sumf;
  }
}

template <int FMT, int R>
__global__ __launch_bounds__(256) void k_gemv_exact_blk(const i32x4* __restrict__ wq, const unsigned short* __restrict__ wd,
                                                        typename ActOf<FMT>::type act, float* __restrict__ out, const float* add, int m, int nb) {
  extern __shared__ __attribute__((aligned(16))) float exact_terms[];
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int wave = blockIdx.x * (blockDim.x >> 6) + wv;
  const int row0 = wave * R;
  if (row0 >= m) return;
  const int nt = (nb + 3) & ~3;  // row stride of the term table (16-byte aligned rows)
  float* T = exact_terms + (size_t)wv * R * nt;
  rows_terms<FMT, R>(wq, wd, act, row0, m, nb, lane, T, nt);
  exact_chain_store<R>(T, nb, nt, row0, m, lane, out, add);
}

template <class P, int R>
__global__ __launch_bounds__(256) void k_gemv_exact_pieces(const char* __restrict__ w, size_t off, size_t n, typename P::Act act,
                                                           float* __restrict__ out, const float* add, int m, int nbr) {
  extern __shared__ __attribute__((aligned(16))) float exact_terms[];
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int wave = blockIdx.x * (blockDim.x >> 6) + wv;
  const int row0 = wave * R;
  if (row0 >= m) return;
  const int nt = (nbr + 3) & ~3;
  float* T = exact_terms + (size_t)wv * R * nt;
  const int np = nbr * P::PIECES;
  for (int c0 = 0; c0 < np; c0 += 64) {
    const int c = c0 + lane;
    const bool live = c < np;
    const int cc = live ? c : np - 1;
    typename P::W wvv[R];
#pragma unroll
    for (int r = 0; r < R; r++) {
      const int row = row0 + r < m ? row0 + r : m - 1;
      wvv[r] = P::load(w, off, n, (size_t)row * nbr, cc);
    }
    const typename P::X x = P::loadx(act, cc);
#pragma unroll
    for (int r = 0; r < R; r++) {
      float t;
      if constexpr (P::PIECES == 1)
        t = P::term(wvv[r], x, lane);
      else
        t = P::term(wvv[r], x, cc, lane);  // the block's one term, on the first lane of its quad
      if (live && (P::PIECES == 1 || (lane & (P::PIECES - 1)) == 0)) T[r * nt + cc / P::PIECES] = t;
    }
  }
  exact_chain_store<R>(T, nbr, nt, row0, m, lane, out, add);
}

template <int R>
__global__ __launch_bounds__(256) void k_gemv_exact_q8k(const i32x4* __restrict__ wq, const float* __restrict__ wd, ActQ8_K act,
                                                        float* __restrict__ out, const float* add, int m, int nsb) {
  extern __shared__ __attribute__((aligned(16))) float exact_terms[];
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int wave = blockIdx.x * (blockDim.x >> 6) + wv;
  const int row0 = wave * R;
  if (row0 >= m) return;
  const int nt = (nsb + 3) & ~3;
  float* T = exact_terms + (size_t)wv * R * nt;
  const int ngroups = nsb * 8;
  for (int g0 = 0; g0 < ngroups; g0 += 64) {
    const int g = g0 + lane;
    const bool live = g < ngroups;
    const int gg = live ? g : ngroups - 1;
    const int sb = gg >> 3;
    const i32x4 x0 = act.q[2 * (size_t)gg], x1 = act.q[2 * (size_t)gg + 1];
    const float d8 = act.d[sb];
#pragma unroll
    for (int r = 0; r < R; r++) {
      const int row = row0 + r < m ? row0 + r : m - 1;
      const size_t gi = (size_t)row * ngroups + gg;
      const i32x4 q0 = __builtin_nontemporal_load(wq + 2 * gi), q1 = __builtin_nontemporal_load(wq + 2 * gi + 1);
      int si = live ? dot_i8x32(q0, q1, x0, x1) : 0;
      si += dpp_i<0xB1>(si);   // the eight 32-element groups of a super-block: exact integer sum (buf_q8_k.rs:216-220)
      si += dpp_i<0x4E>(si);
      si += dpp_i<0x141>(si);  // row_half_mirror: lanes 0-3 <-> 7-4 of each 8
      if (live && (lane & 7) == 0) T[r * nt + sb] = ((float)si * wd[(size_t)row * nsb + sb]) * d8;
    }
  }
  exact_chain_store<R>(T, nsb, nt, row0, m, lane, out, add);
}

// ---- Q4_K / Q5_K in the reference's order at streaming speed ---------------------------------------------------------------------------
// buf_q4_k.rs:192-277 / buf_q5_k.rs:229-325 keep EIGHT f32 lanes per row: inside a super-block `aux32[l] += scale * (q8 * q4)` for the
// elements e with e % 8 == l -- sums of integers below 2^24, exact in f32 in any order --, then per super-block
// `sums[l] += d * aux32[l]` and `sumf -= dmin * sumi`, and at the end `sumf += sums[0..8)` in order.  So a super-block contributes nine
// f32 terms per row: d * A[l] with A[l] the exact integer lane sum, and dmin * sumi.  A lane takes one 16-byte piece of quants as the
// fast kernel does, splits its v_dot4 sums by byte position (byte k of dword i is element class 4 (i & 1) + k: the activation dword
// is masked to one byte per v_dot4), the eight lanes of a super-block add their integers (DPP), and the first of them parks the nine
// terms in LDS; one lane per row then runs the nine chains over the super-blocks in order.  Bit-identical to k_gemv_strict's case.
template <bool Q5, int R>
__global__ __launch_bounds__(256) void k_gemv_exact_q4k(const char* __restrict__ w, size_t off_scale, ActQ8_K act, float* __restrict__ out, const float* add, int m,
                                                        int nsb) {
  extern __shared__ __attribute__((aligned(16))) float exact_terms[];
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int wave = blockIdx.x * (blockDim.x >> 6) + wv;
  const int row0 = wave * R;
  if (row0 >= m) return;
  const size_t n5 = off_scale / 128;  // (Q5_K: blocks in the tensor; planes qs | qh | hdr)
  const i32x4* wq = (const i32x4*)w;
  const i32x4* wqh = (const i32x4*)(w + off_scale);
  const i32x4* wh = (const i32x4*)(w + off_scale + (Q5 ? n5 * 32 : 0));
  const int stride = nsb * 12;  // nine terms per super-block, padded to 12 floats (16-byte aligned records)
  float* T = exact_terms + (size_t)wv * R * stride;
  const int np = nsb * 8;
  if constexpr (!Q5) {
    // Q4_K: class-major planes -- one v_dot4 per class and nibble half (q4k_class_terms, gemv_core.hpp)
    rows_terms_q4k<R, true>(wq, wh, act, row0, m, nsb, lane, T, stride);
    __builtin_amdgcn_s_waitcnt(0xC07F);
    __builtin_amdgcn_wave_barrier();
    if (lane < R && row0 + lane < m) {
      const float sumf = q4k_ordered_sum(T + (size_t)lane * stride, nsb);
      out[row0 + lane] = add ? sumf + add[row0 + lane] : sumf;
    }
    return;
  }
  // Q5_K (planes in the file's element order): the activation dwords masked to one byte per v_dot4
  for (int c0 = 0; c0 < np; c0 += 64) {
    const int c = c0 + lane;
    const bool live = c < np;
    const int cc = live ? c : np - 1;
    const int sb = cc >> 3, j = cc & 7, p = j >> 1, h = j & 1;
    const Q4KX x = q4k_loadx<false>(act, cc);
    // the activation dwords masked to one byte each: xm[i][k] keeps byte k of dword i (class 4 (i & 1) + k)
    int xlm[4][4], xhm[4][4];
#pragma unroll
    for (int i = 0; i < 4; i++)
#pragma unroll
      for (int k = 0; k < 4; k++) {
        xlm[i][k] = (int)((unsigned)x.xl[i] & (0xFFu << (8 * k)));
        xhm[i][k] = (int)((unsigned)x.xh[i] & (0xFFu << (8 * k)));
      }
#pragma unroll
    for (int r = 0; r < R; r++) {
      const size_t blk = (size_t)(row0 + r < m ? row0 + r : m - 1) * nsb + sb;
      const i32x4 qv = __builtin_nontemporal_load(wq + blk * 8 + j);
      const i32x4 hd = __builtin_nontemporal_load(wh + blk);
      i32x4 hv = {0, 0, 0, 0};
      if constexpr (Q5) hv = __builtin_nontemporal_load(wqh + blk * 2 + h);
      const unsigned f = q4k_pair_field((unsigned)hd[1], (unsigned)hd[2], (unsigned)hd[3], p);
      const int sc_lo = (int)(f & 63u), sc_hi = (int)((f >> 6) & 63u);
      const int m_lo = (int)((f >> 12) & 63u), m_hi = (int)(f >> 18);
      int lo[8], hi[8];
#pragma unroll
      for (int l = 0; l < 8; l++) lo[l] = hi[l] = 0;
#pragma unroll
      for (int i = 0; i < 4; i++) {
        const unsigned q = (unsigned)qv[i], hb = (unsigned)hv[i] >> (2 * p);
        const unsigned l4 = (q & 0x0F0F0F0Fu) | (Q5 ? (hb & 0x01010101u) << 4 : 0u);
        const unsigned h4 = ((q >> 4) & 0x0F0F0F0Fu) | (Q5 ? ((hb >> 1) & 0x01010101u) << 4 : 0u);
#pragma unroll
        for (int k = 0; k < 4; k++) {
          lo[4 * (i & 1) + k] = __builtin_amdgcn_sdot4((int)l4, xlm[i][k], lo[4 * (i & 1) + k], false);
          hi[4 * (i & 1) + k] = __builtin_amdgcn_sdot4((int)h4, xhm[i][k], hi[4 * (i & 1) + k], false);
        }
      }
      int A[9];
#pragma unroll
      for (int l = 0; l < 8; l++) A[l] = live ? sc_lo * lo[l] + sc_hi * hi[l] : 0;
      A[8] = live ? m_lo * x.bs_lo + m_hi * x.bs_hi : 0;
#pragma unroll
      for (int l = 0; l < 9; l++) {  // the eight pieces of the super-block: exact integer sums
        A[l] += dpp_i<0xB1>(A[l]);
        A[l] += dpp_i<0x4E>(A[l]);
        A[l] += dpp_i<0x141>(A[l]);
      }
      if (live && (lane & 7) == 0) {
        const unsigned h0 = (unsigned)hd[0];
        const float d = h2f((unsigned short)(h0 & 0xffff)) * x.d8, dmin = h2f((unsigned short)(h0 >> 16)) * x.d8;
        float* t = T + (size_t)r * stride + sb * 12;
#pragma unroll
        for (int l = 0; l < 8; l++) t[l] = d * (float)A[l];
        t[8] = dmin * (float)A[8];
      }
    }
  }
  __builtin_amdgcn_s_waitcnt(0xC07F);
  __builtin_amdgcn_wave_barrier();
  if (lane < R && row0 + lane < m) {
    const float* t = T + (size_t)lane * stride;
    float sums[8], sumf = 0.0f;
#pragma unroll
    for (int l = 0; l < 8; l++) sums[l] = 0.0f;
    for (int sb = 0; sb < nsb; sb++) {
      const f32x4 a = *(const f32x4*)(t + sb * 12), b = *(const f32x4*)(t + sb * 12 + 4);
      sums[0] += a[0];
      sums[1] += a[1];
      sums[2] += a[2];
      sums[3] += a[3];
      sums[4] += b[0];
      sums[5] += b[1];
      sums[6] += b[2];
      sums[7] += b[3];
      sumf -= t[sb * 12 + 8];
    }
#pragma unroll
    for (int l = 0; l < 8; l++) sumf += sums[l];
    out[row0 + lane] = add ? sumf + add[row0 + lane] : sumf;
  }
}

// ---- Q6_K (buf_q6_k.rs:183-234): eight f32 lanes as above, no minimum term; the levels are made signed bytes (q6 - 32) so that the
// byte-split v_dot4 sums are the reference's products directly.  planes ql | qh | scales | d (common.hpp), n = off_scale / 128 blocks.
template <int R>
__global__ __launch_bounds__(256) void k_gemv_exact_q6k(const char* __restrict__ w, size_t off_qh, ActQ8_K act, float* __restrict__ out, const float* add, int m,
                                                        int nsb) {
  extern __shared__ __attribute__((aligned(16))) float exact_terms[];
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int wave = blockIdx.x * (blockDim.x >> 6) + wv;
  const int row0 = wave * R;
  if (row0 >= m) return;
  const size_t n = off_qh / 128;
  const i32x4* wql = (const i32x4*)w;
  const i32x4* wqh = (const i32x4*)(w + off_qh);
  const i32x4* wsc = (const i32x4*)(w + off_qh + n * 64);
  const unsigned short* wd = (const unsigned short*)(w + off_qh + n * 80);
  const int stride = nsb * 8;
  float* T = exact_terms + (size_t)wv * R * stride;
  const int np = nsb * 8;
  for (int c0 = 0; c0 < np; c0 += 64) {
    const int c = c0 + lane;
    const bool live = c < np;
    const int cc = live ? c : np - 1;
    const int sb = cc >> 3, h = (cc >> 2) & 1, a = (cc >> 1) & 1, p = cc & 1;
    const int gi = 8 * h + p + 2 * a;  // the low nibbles' 16-element scale group; the high nibbles' is gi + 4
    const i32x4* xq = act.q + (size_t)sb * 16 + gi;
    const i32x4 xl = xq[0], xh = xq[4];
    const float d8 = act.d[sb];
    int xlm[4][4], xhm[4][4];
#pragma unroll
    for (int i = 0; i < 4; i++)
#pragma unroll
      for (int k = 0; k < 4; k++) {
        xlm[i][k] = (int)((unsigned)xl[i] & (0xFFu << (8 * k)));
        xhm[i][k] = (int)((unsigned)xh[i] & (0xFFu << (8 * k)));
      }
#pragma unroll
    for (int r = 0; r < R; r++) {
      const size_t blk = (size_t)(row0 + r < m ? row0 + r : m - 1) * nsb + sb;
      const i32x4 qv = __builtin_nontemporal_load(wql + blk * 8 + (cc & 7));
      const i32x4 hv = __builtin_nontemporal_load(wqh + blk * 4 + 2 * h + p);
      const i32x4 sc4 = __builtin_nontemporal_load(wsc + blk);
      const int sc_lo = (int)(signed char)(((unsigned)sc4[gi >> 2] >> (8 * (gi & 3))) & 0xFFu);
      const int sc_hi = (int)(signed char)(((unsigned)sc4[(gi + 4) >> 2] >> (8 * (gi & 3))) & 0xFFu);
      int lo[8], hi[8];
#pragma unroll
      for (int l = 0; l < 8; l++) lo[l] = hi[l] = 0;
#pragma unroll
      for (int i = 0; i < 4; i++) {
        const unsigned q = (unsigned)qv[i], hb = (unsigned)hv[i] >> (2 * a);
        const unsigned l6 = (q & 0x0F0F0F0Fu) | ((hb & 0x03030303u) << 4);
        const unsigned h6 = ((q >> 4) & 0x0F0F0F0Fu) | (((hb >> 4) & 0x03030303u) << 4);
        // 0 .. 63 -> signed bytes v - 32 without carries between bytes
        const int ls = (int)((((l6 | 0x80808080u) - 0x20202020u)) ^ 0x80808080u);
        const int hs = (int)((((h6 | 0x80808080u) - 0x20202020u)) ^ 0x80808080u);
#pragma unroll
        for (int k = 0; k < 4; k++) {
          lo[4 * (i & 1) + k] = __builtin_amdgcn_sdot4(ls, xlm[i][k], lo[4 * (i & 1) + k], false);
          hi[4 * (i & 1) + k] = __builtin_amdgcn_sdot4(hs, xhm[i][k], hi[4 * (i & 1) + k], false);
        }
      }
      int A[8];
#pragma unroll
      for (int l = 0; l < 8; l++) {
        A[l] = live ? sc_lo * lo[l] + sc_hi * hi[l] : 0;
        A[l] += dpp_i<0xB1>(A[l]);
        A[l] += dpp_i<0x4E>(A[l]);
        A[l] += dpp_i<0x141>(A[l]);
      }
      if (live && (lane & 7) == 0) {
        const float d = h2f(wd[blk]) * d8;
        float* t = T + (size_t)r * stride + sb * 8;
#pragma unroll
        for (int l = 0; l < 8; l++) t[l] = (float)A[l] * d;
      }
    }
  }
  __builtin_amdgcn_s_waitcnt(0xC07F);
  __builtin_amdgcn_wave_barrier();
  if (lane < R && row0 + lane < m) {
    const float* t = T + (size_t)lane * stride;
    float sums[8], sumf = 0.0f;
#pragma unroll
    for (int l = 0; l < 8; l++) sums[l] = 0.0f;
    for (int sb = 0; sb < nsb; sb++) {
      const f32x4 a = *(const f32x4*)(t + sb * 8), b = *(const f32x4*)(t + sb * 8 + 4);
      sums[0] += a[0];
      sums[1] += a[1];
      sums[2] += a[2];
      sums[3] += a[3];
      sums[4] += b[0];
      sums[5] += b[1];
      sums[6] += b[2];
      sums[7] += b[3];
    }
#pragma unroll
    for (int l = 0; l < 8; l++) sumf += sums[l];
    out[row0 + lane] = add ? sumf + add[row0 + lane] : sumf;
  }
}

// ---- Q3_K (buf_q3_k.rs:238-329): eight i32 lanes per super-block (element e feeds lane e % 8), `sums[l] += d * aux32[l]`, the sums
// reduced in order at the end.  A lane takes one 16-byte qs piece = four 16-element scale groups (PieceQ3_K, gemv_core.hpp); the levels
// (2 low bits | hmask bit << 2) - 4 are made signed bytes and the v_dot4 sums split by byte position as for Q4_K.
template <int R>
__global__ __launch_bounds__(256) void k_gemv_exact_q3k(const char* __restrict__ w, size_t off, size_t n, ActQ8_K act, float* __restrict__ out,
                                                        const float* add, int m, int nsb) {
  extern __shared__ __attribute__((aligned(16))) float exact_terms[];
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int wave = blockIdx.x * (blockDim.x >> 6) + wv;
  const int row0 = wave * R;
  if (row0 >= m) return;
  const int stride = nsb * 8;
  float* T = exact_terms + (size_t)wv * R * stride;
  const int np = nsb * 4;
  for (int c0 = 0; c0 < np; c0 += 64) {
    const int c = c0 + lane;
    const bool live = c < np;
    const int cc = live ? c : np - 1;
    const int sb = cc >> 2;
    const KGroupsX x = kgroups_loadx(act, cc);
#pragma unroll
    for (int r = 0; r < R; r++) {
      const PieceQ3_K::W wv3 = PieceQ3_K::load(w, off, n, (size_t)(row0 + r < m ? row0 + r : m - 1) * nsb, cc);
      int A[8];
#pragma unroll
      for (int l = 0; l < 8; l++) A[l] = 0;
#pragma unroll
      for (int s = 0; s < 4; s++) {
        const int sc = PieceQ3_K::scale(wv3, cc, s);
        const int bit = 4 * ((cc >> 1) & 1) + s;
        int g8[8];
#pragma unroll
        for (int l = 0; l < 8; l++) g8[l] = 0;
#pragma unroll
        for (int i = 0; i < 4; i++) {
          const unsigned u = (((unsigned)wv3.qv[i] >> (2 * s)) & 0x03030303u) | ((((unsigned)wv3.hm[i] >> bit) & 0x01010101u) << 2);
          const int lv = (int)(((u | 0x80808080u) - 0x04040404u) ^ 0x80808080u);  // 0 .. 7 -> signed bytes u - 4
#pragma unroll
          for (int k = 0; k < 4; k++)
            g8[4 * (i & 1) + k] = __builtin_amdgcn_sdot4(lv, (int)((unsigned)x.xq[s][i] & (0xFFu << (8 * k))), g8[4 * (i & 1) + k], false);
        }
#pragma unroll
        for (int l = 0; l < 8; l++) A[l] += sc * g8[l];
      }
#pragma unroll
      for (int l = 0; l < 8; l++) {
        A[l] = quad_sum_i32(live ? A[l] : 0);  // the four pieces of the super-block
      }
      if (live && (lane & 3) == 0) {
        const float d = h2f((unsigned short)((unsigned)wv3.sd[3] & 0xffffu)) * x.d8;
        float* t = T + (size_t)r * stride + sb * 8;
#pragma unroll
        for (int l = 0; l < 8; l++) t[l] = d * (float)A[l];
      }
    }
  }
  __builtin_amdgcn_s_waitcnt(0xC07F);
  __builtin_amdgcn_wave_barrier();
  if (lane < R && row0 + lane < m) {
    const float* t = T + (size_t)lane * stride;
    float sums[8];
#pragma unroll
    for (int l = 0; l < 8; l++) sums[l] = 0.0f;
    for (int sb = 0; sb < nsb; sb++) {
      const f32x4 a = *(const f32x4*)(t + sb * 8), b = *(const f32x4*)(t + sb * 8 + 4);
      sums[0] += a[0];
      sums[1] += a[1];
      sums[2] += a[2];
      sums[3] += a[3];
      sums[4] += b[0];
      sums[5] += b[1];
      sums[6] += b[2];
      sums[7] += b[3];
    }
    float sumf = sums[0];
#pragma unroll
    for (int l = 1; l < 8; l++) sumf = sumf + sums[l];
    out[row0 + lane] = add ? sumf + add[row0 + lane] : sumf;
  }
}

int launch_gemv_strict(crabml_hip_device* dev, const crabml_hip_buf* w, size_t m, size_t k, const void* act, size_t b,
                       float* out, const float* add_all) {
  const uint32_t qt = vec_dot_rhs_dtype(w->dtype);
  if (qt == 0xffffffffu) return set_error(dev, CRABML_HIP_TENSOR_ERROR, "matmul_vec: unsupported weight dtype %u", w->dtype);
  const ActLayout al = act_layout(qt, k);
  const size_t act_stride = qt == CRABML_HIP_F32 ? k * 4 : al.total;
  constexpr int R = 2, WAVES = 4;
  const char* wp = (const char*)w->ptr;
  const size_t nterms = k / block_elems(w->dtype);
  const size_t lds = (size_t)WAVES * R * ((nterms + 3) & ~(size_t)3) * sizeof(float);
  const unsigned grid = (unsigned)((m + R * WAVES - 1) / (R * WAVES));
  // CRABML_HIP_TEST_HOOKS=1 CRABML_HIP_STRICT_SCALAR=1: every format through the one-thread-per-row kernel (the A/B of the tests)
  static const bool scalar_only = [] {
    const char* h = getenv("CRABML_HIP_TEST_HOOKS");
    const char* e = getenv("CRABML_HIP_STRICT_SCALAR");
    return h && h[0] == '1' && e && e[0] == '1';
  }();
  for (size_t bi = 0; bi < b; bi++) {
    const char* ap = (const char*)act + bi * act_stride;
    float* o = out + bi * m;
    const float* add = add_all ? add_all + bi * m : nullptr;  // the residual row of THIS batch row
    bool done = false;
    if (!scalar_only && lds <= 48 * 1024 && block_elems(w->dtype) > 1) {
      done = true;
      const ActQ8_0 a0{(const i32x4*)ap, (const unsigned short*)(ap + al.off_d), (const int*)(ap + al.off_aux)};
      const ActQ8_1 a1{(const i32x4*)ap, (const unsigned short*)(ap + al.off_d), (const unsigned short*)(ap + al.off_aux)};
      const ActQ8_K ak = act_q8k_at(ap, al.off_d, al.off_aux, al.off_p);
      switch (w->dtype) {
        case CRABML_HIP_Q4_0:
          k_gemv_exact_blk<CRABML_HIP_Q4_0, R><<<grid, 64 * WAVES, lds, dev->stream>>>((const i32x4*)wp, (const unsigned short*)(wp + w->wl.off_scale), a0, o, add, (int)m, (int)(k / 32));
          break;
        case CRABML_HIP_Q8_0:
          k_gemv_exact_blk<CRABML_HIP_Q8_0, R><<<grid, 64 * WAVES, lds, dev->stream>>>((const i32x4*)wp, (const unsigned short*)(wp + w->wl.off_scale), a0, o, add, (int)m, (int)(k / 32));
          break;
        case CRABML_HIP_Q4_1:
          k_gemv_exact_blk<CRABML_HIP_Q4_1, R><<<grid, 64 * WAVES, lds, dev->stream>>>((const i32x4*)wp, (const unsigned short*)(wp + w->wl.off_scale), a1, o, add, (int)m, (int)(k / 32));
          break;
        case CRABML_HIP_Q5_0:
          k_gemv_exact_pieces<PieceQ5_0, R><<<grid, 64 * WAVES, lds, dev->stream>>>(wp, w->wl.off_scale, w->wl.n_blocks, a0, o, add, (int)m, (int)(k / 32));
          break;
        case CRABML_HIP_Q5_1:
          k_gemv_exact_pieces<PieceQ5_1, R><<<grid, 64 * WAVES, lds, dev->stream>>>(wp, w->wl.off_scale, w->wl.n_blocks, a1, o, add, (int)m, (int)(k / 32));
          break;
        case CRABML_HIP_Q2_K:
          k_gemv_exact_pieces<PieceQ2_K, R><<<grid, 64 * WAVES, lds, dev->stream>>>(wp, w->wl.off_scale, w->wl.n_blocks, ak, o, add, (int)m, (int)(k / 256));
          break;
        case CRABML_HIP_Q4_K:
        case CRABML_HIP_Q5_K: {
          const size_t lk = (size_t)WAVES * R * (k / 256) * 12 * sizeof(float);
          if (lk > 48 * 1024) {
            done = false;
          } else if (w->dtype == CRABML_HIP_Q4_K) {
            k_gemv_exact_q4k<false, R><<<grid, 64 * WAVES, lk, dev->stream>>>(wp, w->wl.off_scale, ak, o, add, (int)m, (int)(k / 256));
          } else {
            k_gemv_exact_q4k<true, R><<<grid, 64 * WAVES, lk, dev->stream>>>(wp, w->wl.off_scale, ak, o, add, (int)m, (int)(k / 256));
          }
          break;
        }
        case CRABML_HIP_Q6_K: {
          const size_t lk = (size_t)WAVES * R * (k / 256) * 8 * sizeof(float);
          if (lk > 48 * 1024)
            done = false;
          else
            k_gemv_exact_q6k<R><<<grid, 64 * WAVES, lk, dev->stream>>>(wp, w->wl.off_scale, ak, o, add, (int)m, (int)(k / 256));
          break;
        }
        case CRABML_HIP_Q3_K: {
          const size_t lk = (size_t)WAVES * R * (k / 256) * 8 * sizeof(float);
          if (lk > 48 * 1024)
            done = false;
          else
            k_gemv_exact_q3k<R><<<grid, 64 * WAVES, lk, dev->stream>>>(wp, w->wl.off_scale, w->wl.n_blocks, ak, o, add, (int)m, (int)(k / 256));
          break;
        }
        case CRABML_HIP_Q8_K:
          k_gemv_exact_q8k<R><<<grid, 64 * WAVES, lds, dev->stream>>>((const i32x4*)wp, (const float*)(wp + w->wl.off_scale), ak, o, add, (int)m, (int)(k / 256));
          break;
        default: done = false;
      }
    }
    if (!done)
      k_gemv_strict<<<(unsigned)((m + 63) / 64), 64, 0, dev->stream>>>(wp, (int)w->dtype, w->wl.off_scale, ap, al.off_d, al.off_aux, o, add, (int)m, (int)k);
  }
  return 0;
}

}  // namespace crabml_hip

// gemv_strict.hip -- strict-order matmul_vec (CRABML_HIP_FLAG_STRICT_ORDER): a parity instrument.
//
// One thread per output row walks the row's blocks in order and evaluates exactly the reference's
// scalar loops (vec_dot_*_fallback: buf_q4_0.rs:240-253, buf_q8_0.rs:275-286, buf_q4_1.rs:266-280,
// buf_q4_k.rs:192-277, buf_q8_k.rs:211-224, buf_f32.rs:19-27, buf_f16.rs:83-97) -- same integer sums,
// same f32 expression, same association -- on the same device planes the fast kernels read.  Logits are
// then bit-identical to the default (non-SIMD) build of the reference, which turns the end-to-end parity
// check into an equality test.  Not a performance path (uncoalesced by construction).
#include "devutil.hpp"
#include "kernels.hpp"

namespace crabml_hip {

__global__ __launch_bounds__(64) void k_gemv_strict(const char* __restrict__ w, int dtype, size_t off_scale,
                                                    const char* __restrict__ act, size_t off_d, size_t off_aux,
                                                    float* __restrict__ out, int m, int k) {
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
              unsigned char qb = q4[32 * c + e];
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
  out[row] = sumf;
}

int launch_gemv_strict(crabml_hip_device* dev, const crabml_hip_buf* w, size_t m, size_t k, const void* act, size_t b,
                       float* out) {
  const uint32_t qt = vec_dot_rhs_dtype(w->dtype);
  if (qt == 0xffffffffu) return set_error(dev, CRABML_HIP_TENSOR_ERROR, "matmul_vec: unsupported weight dtype %u", w->dtype);
  const ActLayout al = act_layout(qt, k);
  const size_t act_stride = qt == CRABML_HIP_F32 ? k * 4 : al.total;
  for (size_t bi = 0; bi < b; bi++)
    k_gemv_strict<<<(unsigned)((m + 63) / 64), 64, 0, dev->stream>>>((const char*)w->ptr, (int)w->dtype, w->wl.off_scale,
                                                                     (const char*)act + bi * act_stride, al.off_d,
                                                                     al.off_aux, out + bi * m, (int)m, (int)k);
  return 0;
}

}  // namespace crabml_hip

// dequant.hpp -- one element of a device-resident (planes) tensor as f32, with the reference's exact
// expression (one or two roundings): BlockQ*::dequantize in buf_q8_0.rs:18-23, buf_q4_0.rs:18-27,
// buf_q4_1.rs:19-30 (interleaved order, as written there), buf_q4_k.rs:24-47, buf_q8_k.rs:15-20.
#pragma once
#include "devutil.hpp"

namespace crabml_hip {

__device__ __forceinline__ float dequant_elem(const char* __restrict__ w, int dtype, size_t off_scale, size_t e) {
  float v;
  switch (dtype) {
    case CRABML_HIP_F32: v = ((const float*)w)[e]; break;
    case CRABML_HIP_F16: v = h2f(((const unsigned short*)w)[e]); break;
    case CRABML_HIP_Q8_0: {
      size_t b = e / 32;
      float d = h2f(((const unsigned short*)(w + off_scale))[b]);
      v = (float)((const signed char*)w)[e] * d;
      break;
    }
    case CRABML_HIP_Q4_0: {
      size_t b = e / 32, j = e % 32;
      float d = h2f(((const unsigned short*)(w + off_scale))[b]);
      unsigned char q = ((const unsigned char*)w)[b * 16 + (j & 15)];
      int xi = (j < 16 ? (q & 0x0F) : (q >> 4)) - 8;
      v = (float)xi * d;
      break;
    }
    case CRABML_HIP_Q4_1: {  // interleaved order, exactly as buf_q4_1.rs:23-29
      size_t b = e / 32, j = e % 32;
      unsigned dm = ((const unsigned*)(w + off_scale))[b];
      float d = h2f((unsigned short)(dm & 0xffffu)), m = h2f((unsigned short)(dm >> 16));
      unsigned char q = ((const unsigned char*)w)[b * 16 + (j >> 1)];
      float xf = (float)((j & 1) ? ((q >> 4) & 0x0F) : (q & 0x0F));
      v = xf * d + m;
      break;
    }
    case CRABML_HIP_Q4_K: {
      size_t sb = e / 256, j = e % 256;
      const unsigned char* hdr = (const unsigned char*)w + off_scale + sb * 16;
      const unsigned char* qs = (const unsigned char*)w + sb * 128;
      unsigned short dh, mh;
      __builtin_memcpy(&dh, hdr, 2);
      __builtin_memcpy(&mh, hdr + 2, 2);
      float d = h2f(dh), mn = h2f(mh);
      int c = (int)(j / 64), l = (int)(j % 64);
      unsigned u0, u1, u2;
      __builtin_memcpy(&u0, hdr + 4, 4);
      __builtin_memcpy(&u1, hdr + 8, 4);
      __builtin_memcpy(&u2, hdr + 12, 4);
      const unsigned f = q4k_pair_field(u0, u1, u2, c);
      const int s6 = (int)((l >= 32 ? f >> 6 : f) & 63u), m6 = (int)((l >= 32 ? f >> 18 : f >> 12) & 63u);
      float d1 = d * (float)s6, m1 = mn * (float)m6;
      unsigned char q = qs[32 * c + q4k_perm_index(l & 31)];  // (class-major qs plane, common.hpp)
      float qf = (float)(l >= 32 ? (q >> 4) : (q & 0xF));
      v = d1 * qf - m1;
      break;
    }
    case CRABML_HIP_Q5_K: {  // buf_q5_k.rs:24-63; planes qs | qh | hdr with n = off_scale / 128 blocks
      const size_t n = off_scale / 128, sb = e / 256, j = e % 256;
      const unsigned char* hdr = (const unsigned char*)w + off_scale + n * 32 + sb * 16;
      const unsigned char* qs = (const unsigned char*)w + sb * 128;
      const unsigned char* qh = (const unsigned char*)w + off_scale + sb * 32;
      unsigned short dh, mh;
      __builtin_memcpy(&dh, hdr, 2);
      __builtin_memcpy(&mh, hdr + 2, 2);
      float d = h2f(dh), mn = h2f(mh);
      int c = (int)(j / 64), l = (int)(j % 64);
      unsigned u0, u1, u2;
      __builtin_memcpy(&u0, hdr + 4, 4);
      __builtin_memcpy(&u1, hdr + 8, 4);
      __builtin_memcpy(&u2, hdr + 12, 4);
      const unsigned f = q4k_pair_field(u0, u1, u2, c);
      const int s6 = (int)((l >= 32 ? f >> 6 : f) & 63u), m6 = (int)((l >= 32 ? f >> 18 : f >> 12) & 63u);
      float d1 = d * (float)s6, m1 = mn * (float)m6;
      unsigned char q = qs[32 * c + (l & 31)];
      const int hbit = (qh[l & 31] >> (2 * c + (l >= 32 ? 1 : 0))) & 1;
      float qf = (float)(l >= 32 ? (q >> 4) : (q & 0xF)) + (hbit ? 16.0f : 0.0f);
      v = d1 * qf - m1;
      break;
    }
    case CRABML_HIP_Q6_K: {  // buf_q6_k.rs:21-48; planes ql | qh | scales | d with n = off_scale / 128 blocks
      const size_t n = off_scale / 128, sb = e / 256;
      const int j = (int)(e % 256), idx = j / 128, r = j % 128, l = r % 32, quarter = r / 32;
      const unsigned char* ql = (const unsigned char*)w + sb * 128 + 64 * idx;
      const unsigned char* qh = (const unsigned char*)w + off_scale + sb * 64 + 32 * idx;
      const signed char* sc = (const signed char*)w + off_scale + n * 64 + sb * 16 + 8 * idx;
      const float d = h2f(((const unsigned short*)(w + off_scale + n * 80))[sb]);
      const unsigned char lo = quarter & 1 ? ql[l + 32] : ql[l];
      const int nib = quarter >= 2 ? (lo >> 4) : (lo & 0xF);
      const int hi2 = (qh[l] >> (2 * quarter)) & 3;
      const int q = (nib | (hi2 << 4)) - 32;
      v = d * (float)sc[l / 16 + 2 * quarter] * (float)q;
      break;
    }
    case CRABML_HIP_Q5_0: {  // buf_q5_0.rs:22-37; planes qs | qh | d with n = off_scale / 16 blocks
      const size_t n = off_scale / 16, b = e / 32, j = e % 32;
      const unsigned qh = ((const unsigned*)(w + off_scale))[b];
      const float d = h2f(((const unsigned short*)(w + off_scale + n * 4))[b]);
      const unsigned char q = ((const unsigned char*)w)[b * 16 + (j & 15)];
      const int xi = (int)((j < 16 ? (q & 0x0F) : (q >> 4)) | (((qh >> j) & 1u) << 4)) - 16;
      v = (float)xi * d;
      break;
    }
    case CRABML_HIP_Q5_1: {  // buf_q5_1.rs:20-36 (element i | element i + 16 per byte: not interleaved)
      const size_t b = e / 32, j = e % 32;
      const unsigned dm = ((const unsigned*)(w + off_scale))[2 * b], qh = ((const unsigned*)(w + off_scale))[2 * b + 1];
      const unsigned char q = ((const unsigned char*)w)[b * 16 + (j & 15)];
      const unsigned xi = (unsigned)(j < 16 ? (q & 0x0F) : (q >> 4)) | (((qh >> j) & 1u) << 4);
      v = (float)xi * h2f((unsigned short)(dm & 0xffffu)) + h2f((unsigned short)(dm >> 16));
      break;
    }
    case CRABML_HIP_Q2_K: {  // buf_q2_k.rs:35-69; planes qs | scales | (d, dmin) with n = off_scale / 64 blocks
      const size_t n = off_scale / 64, sb = e / 256;
      const int j = (int)(e % 256), g = j / 16, half = g >> 3, s = (g & 7) >> 1, h = g & 1;
      const unsigned char sc = ((const unsigned char*)w)[off_scale + sb * 16 + g];
      const unsigned dm = ((const unsigned*)(w + off_scale + n * 16))[sb];
      const unsigned char q = ((const unsigned char*)w)[sb * 64 + 32 * half + 16 * h + (j & 15)];
      const float dl = h2f((unsigned short)(dm & 0xffffu)) * (float)(sc & 0xF), ml = h2f((unsigned short)(dm >> 16)) * (float)(sc >> 4);
      v = dl * (float)((q >> (2 * s)) & 3) - ml;
      break;
    }
    case CRABML_HIP_Q3_K: {  // buf_q3_k.rs:37-88; planes qs | hmask | (scales[12], d) with n = off_scale / 64 blocks
      const size_t n = off_scale / 64, sb = e / 256;
      const int j = (int)(e % 256), g = j / 16, half = g >> 3, s = (g & 7) >> 1, h = g & 1, pos = 16 * h + (j & 15);
      const unsigned char* sd = (const unsigned char*)w + off_scale + n * 32 + sb * 16;
      const int lo = g < 8 ? (sd[g] & 0xF) : (sd[g - 8] >> 4), hi = (sd[8 + (g & 3)] >> (2 * (g >> 2))) & 3;
      unsigned short dh;
      __builtin_memcpy(&dh, sd + 12, 2);
      const float dl = h2f(dh) * (float)((lo | (hi << 4)) - 32);
      const unsigned char q = ((const unsigned char*)w)[sb * 64 + 32 * half + pos];
      const unsigned char hb = ((const unsigned char*)w)[off_scale + sb * 32 + pos];
      v = dl * (float)((int)((q >> (2 * s)) & 3) - (((hb >> (4 * half + s)) & 1) ? 0 : 4));
      break;
    }
    case CRABML_HIP_Q8_K: {
      size_t sb = e / 256;
      float d = ((const float*)(w + off_scale))[sb];
      v = d * (float)((const signed char*)w)[e];
      break;
    }
    default: v = 0.f;
  }
  return v;
}

}  // namespace crabml_hip

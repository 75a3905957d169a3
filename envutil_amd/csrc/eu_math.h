// Device math that must agree bit for bit with what the reference gets from the
// host libm. envutil's portable ("goading") back-end calls std::atan2/atan on
// float lanes (zimt/simd/vector_common.h:203-246), i.e. glibc's atan2f/atanf.
// glibc is not part of /root/reference; the pinned version is the image's
// glibc 2.35, whose float atan/atan2 are the fdlibm-derived
// sysdeps/ieee754/flt-32/{s_atanf.c,e_atan2f.c}: a fixed sequence of float32
// operations with no FMA variant on x86_64 (sysdeps/x86_64/fpu/multiarch has
// ifuncs for sinf/cosf/expf..., none for atanf/atan2f). The sequence is
// restated here operation by operation; the file is compiled with
// -ffp-contract=off so that no mul/add pair is fused. tests/test_device_math.py
// checks the host compilation of this header against the live libm over all
// 2^32 atanf inputs and ~10^9 atan2f pairs.
#ifndef EU_MATH_H
#define EU_MATH_H

#if defined(__HIPCC__)
#define EU_HD __host__ __device__ __forceinline__
#else
#define EU_HD static inline
#endif

#include <stdint.h>

EU_HD uint32_t eu_f2u(float f) { union { float f; uint32_t u; } c; c.f = f; return c.u; }
EU_HD float eu_u2f(uint32_t u) { union { float f; uint32_t u; } c; c.u = u; return c.f; }

// glibc 2.35 flt-32/s_atanf.c
EU_HD float eu_atanf(float x)
{
  const float atanhi[4] = { 4.6364760399e-01f, 7.8539812565e-01f,
                            9.8279368877e-01f, 1.5707962513e+00f };
  const float atanlo[4] = { 5.0121582440e-09f, 3.7748947079e-08f,
                            3.4473217170e-08f, 7.5497894159e-08f };
  const float aT0 = 3.3333334327e-01f, aT1 = -2.0000000298e-01f,
              aT2 = 1.4285714924e-01f, aT3 = -1.1111110449e-01f,
              aT4 = 9.0908870101e-02f, aT5 = -7.6918758452e-02f,
              aT6 = 6.6610731184e-02f, aT7 = -5.8335702866e-02f,
              aT8 = 4.9768779427e-02f, aT9 = -3.6531571299e-02f,
              aT10 = 1.6285819933e-02f;
  int32_t hx = (int32_t)eu_f2u(x);
  int32_t ix = hx & 0x7fffffff;
  int id;
  if (ix >= 0x4c000000) {              /* |x| >= 2^25 */
    if (ix > 0x7f800000) return x + x; /* NaN */
    if (hx > 0) return atanhi[3] + atanlo[3];
    return -atanhi[3] - atanlo[3];
  }
  if (ix < 0x3ee00000) {               /* |x| < 0.4375 */
    if (ix < 0x31000000) return x;     /* |x| < 2^-29 */
    id = -1;
  } else {
    x = eu_u2f((uint32_t)ix);          /* fabsf */
    if (ix < 0x3f980000) {             /* |x| < 1.1875 */
      if (ix < 0x3f300000) { id = 0; x = (2.0f * x - 1.0f) / (2.0f + x); }
      else                 { id = 1; x = (x - 1.0f) / (x + 1.0f); }
    } else {
      if (ix < 0x401c0000) { id = 2; x = (x - 1.5f) / (1.0f + 1.5f * x); }
      else                 { id = 3; x = -1.0f / x; }
    }
  }
  float z = x * x;
  float w = z * z;
  float s1 = z * (aT0 + w * (aT2 + w * (aT4 + w * (aT6 + w * (aT8 + w * aT10)))));
  float s2 = w * (aT1 + w * (aT3 + w * (aT5 + w * (aT7 + w * aT9))));
  if (id < 0) return x - x * (s1 + s2);
  float hi = id == 0 ? atanhi[0] : id == 1 ? atanhi[1] : id == 2 ? atanhi[2] : atanhi[3];
  float lo = id == 0 ? atanlo[0] : id == 1 ? atanlo[1] : id == 2 ? atanlo[2] : atanlo[3];
  z = hi - ((x * (s1 + s2) - lo) - x);
  return (hx < 0) ? -z : z;
}

// glibc 2.35 flt-32/e_atan2f.c
EU_HD float eu_atan2f(float y, float x)
{
  const float tiny = 1.0e-30f, pi_o_4 = 7.8539818525e-01f,
              pi_o_2 = 1.5707963705e+00f, pi = 3.1415927410e+00f,
              pi_lo = -8.7422776573e-08f;
  int32_t hx = (int32_t)eu_f2u(x), hy = (int32_t)eu_f2u(y);
  int32_t ix = hx & 0x7fffffff, iy = hy & 0x7fffffff;
  if (ix > 0x7f800000 || iy > 0x7f800000) return x + y;   /* NaN */
  if (hx == 0x3f800000) return eu_atanf(y);               /* x = 1.0 */
  int m = ((hy >> 31) & 1) | ((hx >> 30) & 2);            /* 2*sign(x)+sign(y) */
  if (iy == 0) {
    switch (m) {
      case 0:
      case 1: return y;
      case 2: return pi + tiny;
      default: return -pi - tiny;
    }
  }
  if (ix == 0) return (hy < 0) ? -pi_o_2 - tiny : pi_o_2 + tiny;
  if (ix == 0x7f800000) {
    if (iy == 0x7f800000) {
      switch (m) {
        case 0: return pi_o_4 + tiny;
        case 1: return -pi_o_4 - tiny;
        case 2: return 3.0f * pi_o_4 + tiny;
        default: return -3.0f * pi_o_4 - tiny;
      }
    } else {
      switch (m) {
        case 0: return 0.0f;
        case 1: return -0.0f;
        case 2: return pi + tiny;
        default: return -pi - tiny;
      }
    }
  }
  if (iy == 0x7f800000) return (hy < 0) ? -pi_o_2 - tiny : pi_o_2 + tiny;
  int32_t k = (iy - ix) >> 23;
  float z;
  if (k > 60) z = pi_o_2 + 0.5f * pi_lo;                  /* |y/x| > 2^60 */
  else if (hx < 0 && k < -60) z = 0.0f;                   /* |y|/x < -2^60 */
  else {
    float q = y / x;
    z = eu_atanf(eu_u2f(eu_f2u(q) & 0x7fffffffu));
  }
  switch (m) {
    case 0: return z;
    case 1: return eu_u2f(eu_f2u(z) ^ 0x80000000u);
    case 2: return pi - (z - pi_lo);
    default: return (z - pi_lo) - pi;
  }
}

#endif

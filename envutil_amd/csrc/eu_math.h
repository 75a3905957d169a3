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

// ---------------------------------------------------------------------------
// sinf / cosf: glibc 2.35 sysdeps/ieee754/flt-32/{s_sinf.c,s_cosf.c,sincosf.h}
// (the ARM optimized-routines code: double-precision polynomials). On x86_64
// glibc selects, by ifunc, a variant compiled with -mfma on every CPU that has
// FMA (sysdeps/x86_64/fpu/multiarch/s_sinf-fma.c); gcc contracts a*b+c there.
// The contraction pattern below is the one in the image's libm.so.6
// (__sinf_fma / __cosf_fma, read from its disassembly; constants read from its
// __sincosf_table and __inv_pio4). Hosts without FMA run a different variant
// whose results can differ in the last bit: the parity tests run on FMA hosts.
// ---------------------------------------------------------------------------

#if defined(__HIPCC__) || defined(__FMA__) || defined(EU_MATH_HAVE_FMA)
#define EU_HAVE_SINCOSF 1

EU_HD double eu_fma64(double a, double b, double c) { return __builtin_fma(a, b, c); }

// sinf_poly (sincosf.h): n even -> sine polynomial on (x, x2); n odd -> cosine
// polynomial on x2. neg selects __sincosf_table[1] (negated cosine terms).
EU_HD float eu_sin_poly(double x, double x2)
{
  const double S1 = -0x1.555545995a603p-3, S2 = 0x1.1107605230bc4p-7, S3 = -0x1.994eb3774cf24p-13;
  double x3 = x * x2;
  double s1 = eu_fma64(S3, x2, S2);
  double x7 = x3 * x2;
  double t = eu_fma64(x3, S1, x);
  return (float)eu_fma64(s1, x7, t);
}

EU_HD float eu_cos_poly(double x2, int neg)
{
  double C0 = 0x1p0, C1 = -0x1.ffffffd0c621cp-2, C2 = 0x1.55553e1068f19p-5,
         C3 = -0x1.6c087e89a359dp-10, C4 = 0x1.99343027bf8c3p-16;
  if (neg) { C0 = -C0; C1 = -C1; C2 = -C2; C3 = -C3; C4 = -C4; }
  double x4 = x2 * x2;
  double c1 = eu_fma64(C1, x2, C0);
  double c2 = eu_fma64(C4, x2, C3);
  double x6 = x4 * x2;
  double c = eu_fma64(x4, C2, c1);
  return (float)eu_fma64(c2, x6, c);
}

// reduce_large (sincosf.h) for |x| >= 120
EU_HD double eu_reduce_large(uint32_t xi, int *np)
{
  const uint32_t inv_pio4[24] = {
    0xa2, 0xa2f9, 0xa2f983, 0xa2f9836e, 0xf9836e4e, 0x836e4e44, 0x6e4e4415, 0x4e441529,
    0x441529fc, 0x1529fc27, 0x29fc2757, 0xfc2757d1, 0x2757d1f5, 0x57d1f534, 0xd1f534dd,
    0xf534ddc0, 0x34ddc0db, 0xddc0db62, 0xc0db6295, 0xdb629599, 0x6295993c, 0x95993c43,
    0x993c4390, 0x3c439041 };
  const uint32_t *arr = &inv_pio4[(xi >> 26) & 15];
  int shift = (xi >> 23) & 7;
  uint64_t n, res0, res1, res2;
  xi = (xi & 0xffffff) | 0x800000;
  xi <<= shift;
  res0 = (uint32_t)(xi * arr[0]);
  res1 = (uint64_t)xi * arr[4];
  res2 = (uint64_t)xi * arr[8];
  res0 = (res2 >> 32) | (res0 << 32);
  res0 += res1;
  n = (res0 + (1ULL << 61)) >> 62;
  res0 -= n << 62;
  double x = (double)(int64_t)res0;
  *np = (int)n;
  return x * 0x1.921fb54442d18p-62;
}

// which = 0: sinf, 1: cosf
EU_HD float eu_sincosf_impl(float y, int which)
{
  const double hpi_inv = 0x1.45f306dc9c883p+23, hpi = 0x1.921fb54442d18p+0;
  double x = (double)y;
  uint32_t top = (eu_f2u(y) >> 20) & 0x7ff;
  if (top <= 0x3f3) {                  // |y| < pi/4
    double x2 = x * x;
    if (top <= 0x397) return which ? 1.0f : y;     // |y| < 2^-12
    return which ? eu_cos_poly(x2, 0) : eu_sin_poly(x, x2);
  }
  int n;
  double xr;
  int sign = 0;
  if (top <= 0x42e) {                  // |y| < 120: reduce_fast
    double r = x * hpi_inv;
    n = ((int32_t)r + 0x800000) >> 24;
    xr = eu_fma64(-(double)n, hpi, x);
  } else if (top <= 0x7f7) {
    uint32_t xi = eu_f2u(y);
    sign = (int)(xi >> 31);
    xr = eu_reduce_large(xi, &n);
  } else {
    return (y - y) / (y - y);          // __math_invalidf: NaN
  }
  // sin: s = sign[(n + sign) & 3], table by (n + sign) & 2, polynomial by n
  // cos: s = sign[(n + sign) & 3], table by (n + sign) & 2, polynomial by n ^ 1
  // (for |y| < 120 sign = 0: reduce_fast keeps the sign in x)
  int ns = n + sign;
  double s = ((ns & 3) == 1 || (ns & 3) == 2) ? -1.0 : 1.0;
  int neg = (ns & 2) != 0;
  int odd = (n ^ which) & 1;
  double x2 = xr * xr;
  if (!odd) return eu_sin_poly(xr * s, x2);
  return eu_cos_poly(x2, neg);
}

/* glibc 2.35 tanf (sysdeps/ieee754/flt-32/s_tanf.c, k_tanf.c, e_rem_pio2f.c: fdlibm's float code, no FMA
 * variant) for the arguments ba6_to_ray_t produces (geometry.h:855-1000: in-face coordinate * pi/4): no
 * reduction up to pi/4, the n = +-1 special case of rem_pio2f up to 3 pi/4. Host check against the live
 * libm, every float: identical for |x| <= 1.375 (in-face coordinates up to 1.75); in (1.387, 2.356) 963 of
 * 2.1e8 arguments differ by one ulp (most of them next to pi/2), beyond 3 pi/4 it returns NaN. */
EU_HD float eu_ktanf(float x, float y, int iy)
{
  const float T0 = 3.3333334327e-01f, T1 = 1.3333334029e-01f, T2 = 5.3968254477e-02f, T3 = 2.1869488060e-02f,
              T4 = 8.8632395491e-03f, T5 = 3.5920790397e-03f, T6 = 1.4562094584e-03f, T7 = 5.8804126456e-04f,
              T8 = 2.4646313977e-04f, T9 = 7.8179444245e-05f, T10 = 7.1407252108e-05f, T11 = -1.8558637748e-05f,
              T12 = 2.5907305826e-05f;
  const float pio4 = 7.8539812565e-01f, pio4lo = 3.7748947079e-08f;
  float z, r, v, w, s;
  const int32_t hx = (int32_t)eu_f2u(x);
  const int32_t ix = hx & 0x7fffffff;
  if (ix < 0x31800000) {
    if ((int)x == 0) {
      if ((ix | (iy + 1)) == 0) return 1.0f / eu_u2f((uint32_t)ix);
      else if (iy == 1) return x;
      else return -1.0f / x;
    }
  }
  if (ix >= 0x3f2ca140) {
    if (hx < 0) { x = -x; y = -y; }
    z = pio4 - x;
    w = pio4lo - y;
    x = z + w; y = 0.0f;
    if (eu_u2f(eu_f2u(x) & 0x7fffffffu) < 0x1p-13f) return (float)((1 - ((hx >> 30) & 2)) * iy) * (1.0f - (float)(2 * iy) * x);
  }
  z = x * x;
  w = z * z;
  r = T1 + w * (T3 + w * (T5 + w * (T7 + w * (T9 + w * T11))));
  v = z * (T2 + w * (T4 + w * (T6 + w * (T8 + w * (T10 + w * T12)))));
  s = z * x;
  r = y + z * (s * (r + v) + y);
  r = r + T0 * s;
  w = x + r;
  if (ix >= 0x3f2ca140) {
    v = (float)iy;
    return (float)(1 - ((hx >> 30) & 2)) * (v - 2.0f * (x - (w * w / (w + v) - r)));
  }
  if (iy == 1) return w;
  {
    float a, t;
    z = eu_u2f(eu_f2u(w) & 0xfffff000u);
    v = r - (z - x);
    t = a = -1.0f / w;
    t = eu_u2f(eu_f2u(t) & 0xfffff000u);
    s = 1.0f + t * z;
    return t + a * (s + t * v);
  }
}

EU_HD float eu_tanf(float x)
{
  const float pio2_1 = 1.5707855225e+00f, pio2_1t = 1.0804334124e-05f, pio2_2 = 1.0804273188e-05f,
              pio2_2t = 6.0770999344e-11f;
  const int32_t hx = (int32_t)eu_f2u(x), ix = hx & 0x7fffffff;
  float y0, y1, z;
  if (ix <= 0x3f490fda) return eu_ktanf(x, 0.0f, 1);
  if (ix >= 0x7f800000) return x - x;
  if (ix < 0x4016cbe4) {
    if (hx > 0) {
      z = x - pio2_1;
      if ((ix & 0xfffffff0) != 0x3fc90fd0) { y0 = z - pio2_1t; y1 = (z - y0) - pio2_1t; }
      else { z = z - pio2_2; y0 = z - pio2_2t; y1 = (z - y0) - pio2_2t; }
    } else {
      z = x + pio2_1;
      if ((ix & 0xfffffff0) != 0x3fc90fd0) { y0 = z + pio2_1t; y1 = (z - y0) + pio2_1t; }
      else { z = z + pio2_2; y0 = z + pio2_2t; y1 = (z - y0) + pio2_2t; }
    }
    return eu_ktanf(y0, y1, -1);
  }
  return eu_u2f(0x7fc00000u);
}

EU_HD float eu_sinf(float y) { return eu_sincosf_impl(y, 0); }
EU_HD float eu_cosf(float y) { return eu_sincosf_impl(y, 1); }

// eu_sinf(y) and eu_cosf(y) for |y| < 120 (the fisheye mounts' phi = atan2f(..) is within +-pi) from ONE
// reduction and one polynomial of each kind, without branches. Same bits as the two calls:
//  * below 0.75 sincosf.h skips the reduction; reduce_fast gives n = 0 and xr = fma(-0.0, hpi, x) = x there,
//    i.e. the same operands for the same polynomials (s = 1, the cosine's constants not negated),
//  * sinf takes the sine polynomial for even n and the cosine polynomial for odd n, cosf the other way round,
//    both on (xr * s, x2, n & 2): the two calls evaluate the same two polynomials and swap them,
//  * negating every constant of the cosine polynomial negates its (rounded) result exactly.
// Lanes of a wavefront differ in n, so the two separate calls ran both polynomials TWICE each.
EU_HD void eu_sincosf_120(float y, float *sn, float *cs)
{
  const double hpi_inv = 0x1.45f306dc9c883p+23, hpi = 0x1.921fb54442d18p+0;
  const double x = (double)y;
  const uint32_t top = (eu_f2u(y) >> 20) & 0x7ff;
  const double r = x * hpi_inv;
  const int n = ((int32_t)r + 0x800000) >> 24;
  const double xr = eu_fma64(-(double)n, hpi, x);
  const double s = ((n & 3) == 1 || (n & 3) == 2) ? -1.0 : 1.0;
  const double x2 = xr * xr;
  const float sv = eu_sin_poly(xr * s, x2);
  float cv = eu_cos_poly(x2, 0);
  if (n & 2) cv = -cv;
  *sn = (n & 1) ? cv : sv;
  *cs = (n & 1) ? sv : cv;
  if (top <= 0x397) { *sn = y; *cs = 1.0f; }   // |y| < 2^-12
}
#endif

// ---------------------------------------------------------------------------
// stereographic_stepper::work (stepper.h:1146): a = M_PI_2 - 2.0 * atan(norm / 2.0)
// is formed in double (libm's double atan) and narrows to float before sincos.
// glibc's double atan (IBM accurate library, table driven) is not restated;
// instead the fdlibm double algorithm below (error < 1 ulp of double) is used and
// the NARROWED float result is compared with the live libm for every float
// norm >= 0 (tests/test_device_math.py::test_stereographic_angle_all_floats):
// the map float -> float is identical on all 2^31 inputs, which is what parity
// needs. IEEE double add/mul/div only, no contraction.
// ---------------------------------------------------------------------------
EU_HD double eu_atan_pos_d(double x)
{
  const double atanhi[4] = { 4.63647609000806093515e-01, 7.85398163397448278999e-01,
                             9.82793723247329054082e-01, 1.57079632679489655800e+00 };
  const double atanlo[4] = { 2.26987774529616870924e-17, 3.06161699786838301793e-17,
                             1.39033110312309984516e-17, 6.12323399573676603587e-17 };
  const double aT0 = 3.33333333333329318027e-01, aT1 = -1.99999999998764832476e-01,
               aT2 = 1.42857142725034663711e-01, aT3 = -1.11111104054623557880e-01,
               aT4 = 9.09088713343650656196e-02, aT5 = -7.69187620504482999495e-02,
               aT6 = 6.66107313738753120669e-02, aT7 = -5.83357013379057348645e-02,
               aT8 = 4.97687799461593236017e-02, aT9 = -3.65315727442169155270e-02,
               aT10 = 1.62858201153657823623e-02;
  int id;
  if (!(x < 7.378697629483821e+19)) {       // x >= 2^66 or NaN
    if (x != x) return x + x;
    return atanhi[3] + atanlo[3];
  }
  if (x < 0.4375) {
    if (x < 7.450580596923828e-09) return x; // 2^-27
    id = -1;
  } else if (x < 1.1875) {
    if (x < 0.6875) { id = 0; x = (2.0 * x - 1.0) / (2.0 + x); }
    else            { id = 1; x = (x - 1.0) / (x + 1.0); }
  } else {
    if (x < 2.4375) { id = 2; x = (x - 1.5) / (1.0 + 1.5 * x); }
    else            { id = 3; x = -1.0 / x; }
  }
  const double z = x * x, w = z * z;
  const double s1 = z * (aT0 + w * (aT2 + w * (aT4 + w * (aT6 + w * (aT8 + w * aT10)))));
  const double s2 = w * (aT1 + w * (aT3 + w * (aT5 + w * (aT7 + w * aT9))));
  if (id < 0) return x - x * (s1 + s2);
  return atanhi[id] - ((x * (s1 + s2) - atanlo[id]) - x);
}

EU_HD float eu_ster_angle(float nrm)
{
  return (float)(1.57079632679489661923 - 2.0 * eu_atan_pos_d((double)nrm / 2.0));
}

#endif

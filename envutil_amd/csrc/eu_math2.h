// Two-pixels-per-lane device math: every arithmetic step works on a float2 so
// that hipcc emits packed fp32 instructions (v_pk_mul_f32 / v_pk_add_f32 /
// v_pk_fma_f32: two IEEE operations per lane per issue slot, measured 4.6 vs
// 4.1 cycles per wave instruction, tools/ubench_valu.hip). Results are the same
// bits as the scalar code in eu_math.h: packing changes how operations are
// issued, never which operations are performed.
//
// Divisions and square roots are the expensive part of the coordinate pipeline
// (measured 47 and 53 cycles per wave for hipcc's correctly rounded fp32 `/`
// and sqrtf). Where the operand range is known, eu_div2_safe / eu_sqrt2_safe
// run the SAME Newton/FMA sequence LLVM lowers `/` and sqrtf to, minus the
// range scaling (v_div_scale, v_div_fmas' scale flag) and the special-value
// fix-up (v_div_fixup), which are the identity for operands in the safe range.
// tests/test_gpu_math.py checks them on the device against `/` and sqrtf.
#ifndef EU_MATH2_H
#define EU_MATH2_H

#include "eu_math.h"

typedef float eu_f2 __attribute__((ext_vector_type(2)));
typedef int eu_i2 __attribute__((ext_vector_type(2)));
typedef unsigned eu_u2 __attribute__((ext_vector_type(2)));

#if defined(__HIPCC__)
#define EU_D2 __host__ __device__ __forceinline__
#else
#define EU_D2 static inline
#endif

EU_D2 eu_f2 eu_fma2(eu_f2 a, eu_f2 b, eu_f2 c)
{
  return __builtin_elementwise_fma(a, b, c);
}
EU_D2 eu_f2 eu_abs2(eu_f2 a) { return __builtin_elementwise_abs(a); }
EU_D2 eu_u2 eu_bits2(eu_f2 a) { return __builtin_bit_cast(eu_u2, a); }
EU_D2 eu_f2 eu_float2(eu_u2 a) { return __builtin_bit_cast(eu_f2, a); }
EU_D2 eu_f2 eu_sel2(eu_i2 m, eu_f2 a, eu_f2 b) { return m ? a : b; }
EU_D2 eu_i2 eu_sel2i(eu_i2 m, eu_i2 a, eu_i2 b) { return m ? a : b; }

// n / d, correctly rounded, for |n| in {0} u [2^-90, 2^90], |d| in [2^-90, 2^90]
// (LLVM AMDGPU LowerFDIV32 without scaling and fix-up)
EU_D2 eu_f2 eu_div2_safe(eu_f2 n, eu_f2 d)
{
#if !defined(__HIP_DEVICE_COMPILE__)
  // host compilation (tests): the device sequence below equals the correctly
  // rounded quotient in the safe range, which is what `/` is on the host
  return n / d;
#else
  eu_f2 r = { __builtin_amdgcn_rcpf(d.x), __builtin_amdgcn_rcpf(d.y) };
  const eu_f2 one = { 1.0f, 1.0f };
  eu_f2 e = eu_fma2(-d, r, one);
  r = eu_fma2(e, r, r);
  eu_f2 q = n * r;
  e = eu_fma2(-d, q, n);
  q = eu_fma2(e, r, q);
  e = eu_fma2(-d, q, n);
  return eu_fma2(e, r, q);
#endif
}

// sqrt(x), correctly rounded, for x in [2^-90, 2^90] (LLVM lowerFSQRTF32's
// rsq + FMA refinement without the scaling of tiny inputs and the 0/inf select)
EU_D2 eu_f2 eu_sqrt2_safe(eu_f2 x)
{
#if !defined(__HIP_DEVICE_COMPILE__)
  return (eu_f2){ __builtin_sqrtf(x.x), __builtin_sqrtf(x.y) };
#else
  eu_f2 r = { __builtin_amdgcn_rsqf(x.x), __builtin_amdgcn_rsqf(x.y) };
  const eu_f2 half = { 0.5f, 0.5f };
  eu_f2 g = x * r;
  eu_f2 h = half * r;
  eu_f2 e = eu_fma2(-h, g, half);
  g = eu_fma2(g, e, g);
  h = eu_fma2(h, e, h);
  eu_f2 dd = eu_fma2(-g, g, x);
  return eu_fma2(dd, h, g);
#endif
}

// x / c for a constant c with rc = RN(1/c): q = x*rc; q' = fma(fma(-q, c, x), rc, q).
// Exact for a given c only where verified (eu_hip checks every float x of the
// range it uses at source creation, eu_setup.hip: verify_const_div_kernel).
EU_D2 eu_f2 eu_div2_const(eu_f2 x, float c, float rc)
{
  const eu_f2 cc = { c, c }, rr = { rc, rc };
  eu_f2 q = x * rr;
  eu_f2 r = eu_fma2(-q, cc, x);
  return eu_fma2(r, rr, q);
}

// atanf for t >= 0 (finite or +inf), both lanes; same bits as eu_atanf.
// One division per lane: the argument reduction of s_atanf.c is num/den with
// (num, den) chosen by range, |t| < 7/16 uses t/1.
EU_D2 eu_f2 eu_atanf_pos2(eu_f2 t)
{
  const eu_u2 it = eu_bits2(t);
  const eu_i2 small = it < 0x3ee00000u, r0 = it < 0x3f300000u, r1 = it < 0x3f980000u,
              r2 = it < 0x401c0000u, big = it >= 0x4c000000u;
  const eu_f2 one = { 1.0f, 1.0f }, two = { 2.0f, 2.0f }, onep5 = { 1.5f, 1.5f },
              mone = { -1.0f, -1.0f };
  // id 0: (2t-1)/(2+t); 1: (t-1)/(t+1); 2: (t-1.5)/(1+1.5t); 3: -1/t; small: t/1
  eu_f2 num = eu_sel2(small, t, eu_sel2(r0, two * t - one, eu_sel2(r1, t - one, eu_sel2(r2, t - onep5, mone))));
  eu_f2 den = eu_sel2(small, one, eu_sel2(r0, two + t, eu_sel2(r1, t + one, eu_sel2(r2, one + onep5 * t, t))));
  // big lanes (t >= 2^25, possibly inf) are overridden below; keep the
  // division in range for them
  den = eu_sel2(big, one, den);
  eu_f2 x = eu_div2_safe(num, den);
  const eu_f2 hi = eu_sel2(r0, (eu_f2){ 4.6364760399e-01f, 4.6364760399e-01f },
                   eu_sel2(r1, (eu_f2){ 7.8539812565e-01f, 7.8539812565e-01f },
                   eu_sel2(r2, (eu_f2){ 9.8279368877e-01f, 9.8279368877e-01f },
                               (eu_f2){ 1.5707962513e+00f, 1.5707962513e+00f })));
  const eu_f2 lo = eu_sel2(r0, (eu_f2){ 5.0121582440e-09f, 5.0121582440e-09f },
                   eu_sel2(r1, (eu_f2){ 3.7748947079e-08f, 3.7748947079e-08f },
                   eu_sel2(r2, (eu_f2){ 3.4473217170e-08f, 3.4473217170e-08f },
                               (eu_f2){ 7.5497894159e-08f, 7.5497894159e-08f })));
  const float aT0 = 3.3333334327e-01f, aT1 = -2.0000000298e-01f, aT2 = 1.4285714924e-01f,
              aT3 = -1.1111110449e-01f, aT4 = 9.0908870101e-02f, aT5 = -7.6918758452e-02f,
              aT6 = 6.6610731184e-02f, aT7 = -5.8335702866e-02f, aT8 = 4.9768779427e-02f,
              aT9 = -3.6531571299e-02f, aT10 = 1.6285819933e-02f;
  eu_f2 z = x * x;
  eu_f2 w = z * z;
  eu_f2 s1 = z * (aT0 + w * (aT2 + w * (aT4 + w * (aT6 + w * (aT8 + w * aT10)))));
  eu_f2 s2 = w * (aT1 + w * (aT3 + w * (aT5 + w * (aT7 + w * aT9))));
  eu_f2 xs = x * (s1 + s2);
  eu_f2 rsmall = x - xs;
  eu_f2 rmid = hi - ((xs - lo) - x);
  eu_f2 r = eu_sel2(small, rsmall, rmid);
  const float hb = 1.5707962513e+00f + 7.5497894159e-08f;   // atanhi[3] + atanlo[3]
  return eu_sel2(big, (eu_f2){ hb, hb }, r);
}

// atan2f(y, x) for both lanes; same bits as eu_atan2f (glibc 2.35 e_atan2f.c).
// Lanes whose operands are outside [2^-40, 2^40] (zero, denormal, huge, inf,
// NaN) take the scalar restatement; for all others e_atan2f.c reduces to
// "z = atanf(|y/x|), then the quadrant fix" (Numerics).
EU_D2 eu_f2 eu_atan2f_2(eu_f2 y, eu_f2 x)
{
  const eu_u2 hx = eu_bits2(x), hy = eu_bits2(y);
  const eu_u2 ix = hx & 0x7fffffffu, iy = hy & 0x7fffffffu;
  // 2^-40 = 0x2b800000, 2^40 = 0x53800000
  const eu_i2 okx = (ix - 0x2b800000u) <= (0x53800000u - 0x2b800000u);
  const eu_i2 oky = (iy - 0x2b800000u) <= (0x53800000u - 0x2b800000u);
  const eu_i2 ok = okx & oky;
  const eu_f2 one = { 1.0f, 1.0f };
  // out-of-range lanes divide 1/1 and are replaced afterwards
  eu_f2 q = eu_div2_safe(eu_sel2(ok, y, one), eu_sel2(ok, x, one));
  eu_f2 z = eu_atanf_pos2(eu_abs2(q));
  const float pi = 3.1415927410e+00f, pi_lo = -8.7422776573e-08f;
  const eu_i2 xneg = (eu_i2)(hx >> 31) != 0, yneg = (eu_i2)(hy >> 31) != 0;
  // m = 0: z; 1: -z; 2: pi - (z - pi_lo); 3: (z - pi_lo) - pi
  eu_f2 zl = z - pi_lo;
  eu_f2 rpos = eu_sel2(yneg, -z, z);
  eu_f2 rneg = eu_sel2(yneg, zl - pi, pi - zl);
  eu_f2 r = eu_sel2(xneg, rneg, rpos);
  if (__builtin_expect(!(ok.x & ok.y), 0)) {
    if (!ok.x) r.x = eu_atan2f(y.x, x.x);
    if (!ok.y) r.y = eu_atan2f(y.y, x.y);
  }
  return r;
}

// atan2f(y, x) for lanes with x > 0 known (x = sqrt(...)): only the sign of y
// selects the quadrant
EU_D2 eu_f2 eu_atan2f_2_xpos(eu_f2 y, eu_f2 x)
{
  const eu_u2 hx = eu_bits2(x), hy = eu_bits2(y);
  const eu_u2 iy = hy & 0x7fffffffu;
  const eu_i2 okx = (hx - 0x2b800000u) <= (0x53800000u - 0x2b800000u);
  const eu_i2 oky = (iy - 0x2b800000u) <= (0x53800000u - 0x2b800000u);
  const eu_i2 ok = okx & oky;
  const eu_f2 one = { 1.0f, 1.0f };
  eu_f2 q = eu_div2_safe(eu_sel2(ok, y, one), eu_sel2(ok, x, one));
  eu_f2 z = eu_atanf_pos2(eu_abs2(q));
  // m = 0: z; m = 1: z with the sign bit flipped
  eu_f2 r = eu_float2(eu_bits2(z) ^ (hy & 0x80000000u));
  if (__builtin_expect(!(ok.x & ok.y), 0)) {
    if (!ok.x) r.x = eu_atan2f(y.x, x.x);
    if (!ok.y) r.y = eu_atan2f(y.y, x.y);
  }
  return r;
}

// ---------------------------------------------------------------------------
// Table-driven variant of eu_atanf_pos2. s_atanf.c's argument reduction is
// x = num/den with (num, den) = (a*t + b, c*t + d) and a per-range (hi, lo):
//   |t| < 7/16          : ( t      ,  1       )          a,b,c,d = 1, 0, 0, 1
//   7/16 <= |t| < 11/16 : ( 2t - 1 ,  2 + t   )                    2,-1, 1, 2
//   11/16 <= |t| < 19/16: ( t - 1  ,  t + 1   )                    1,-1, 1, 1
//   19/16 <= |t| < 39/16: ( t - 1.5,  1 + 1.5t)                    1,-1.5,1.5,1
//   39/16 <= |t|        : ( -1     ,  t       )                    0,-1, 1, 0
// (a*t is exact for a in {0, 1, 2}; RN(RN(c*t) + d) is the reference's own
// expression for every row). All range boundaries are multiples of 2^18 in the
// float's bit pattern, so (bits >> 18) indexes an 81-entry table of
// {a, b, c, d, hi, lo, 0, 0}; the selects of eu_atanf_pos2 become two 16-byte
// reads (LDS on the device). Same bits as eu_atanf / glibc for every t >= 0.
// ---------------------------------------------------------------------------
#define EU_ATAN_TAB_ENTRIES 81
#define EU_ATAN_TAB_FLOATS (EU_ATAN_TAB_ENTRIES * 8)

EU_D2 void eu_atan_tab_entry(int idx, float *e)
{
  // idx = clamp((bits >> 18) - 0xfb7, 0, 80); boundaries 0x3ee00000, 0x3f300000,
  // 0x3f980000, 0x401c0000 >> 18 = 0xfb8, 0xfcc, 0xfe6, 0x1007.
  // Selects on literals, not indexed constant arrays: a kernel prologue fills the table
  // without a trip to memory.
  const int k = idx == 0 ? 0 : idx <= 20 ? 1 : idx <= 46 ? 2 : idx <= 79 ? 3 : 4;
#define EU_TAB5(v0, v1, v2, v3, v4) (k == 0 ? (v0) : k == 1 ? (v1) : k == 2 ? (v2) : k == 3 ? (v3) : (v4))
  e[0] = EU_TAB5(1.0f, 2.0f, 1.0f, 1.0f, 0.0f);
  e[1] = EU_TAB5(0.0f, -1.0f, -1.0f, -1.5f, -1.0f);
  e[2] = EU_TAB5(0.0f, 1.0f, 1.0f, 1.5f, 1.0f);
  e[3] = EU_TAB5(1.0f, 2.0f, 1.0f, 1.0f, 0.0f);
  e[4] = EU_TAB5(0.0f, 4.6364760399e-01f, 7.8539812565e-01f, 9.8279368877e-01f, 1.5707962513e+00f);
  e[5] = EU_TAB5(0.0f, 5.0121582440e-09f, 3.7748947079e-08f, 3.4473217170e-08f, 7.5497894159e-08f);
  e[6] = 0.0f; e[7] = 0.0f;
#undef EU_TAB5
}

EU_D2 eu_f2 eu_atanf_pos2_tab(eu_f2 t, const float *tab)
{
  // round 3: the table fields feed scalar operations (as operands of packed operations they first had
  // to be moved into register pairs: 12 v_mov), num = a * t + b is ONE fma (a in {0, 1, 2}: a * t is exact),
  // and the |t| < 7/16 row (hi = lo = 0) needs no select: 0 - ((xs - 0) - x) is x - xs bit for bit
  const eu_u2 it = eu_bits2(t);
  eu_i2 idx = (eu_i2)(it >> 18) - 0xfb7;
  idx = __builtin_elementwise_min(__builtin_elementwise_max(idx, (eu_i2){ 0, 0 }), (eu_i2){ 80, 80 });
  const float *e0 = tab + idx.x * 8, *e1 = tab + idx.y * 8;
  const float a0 = e0[0], b0 = e0[1], c0 = e0[2], h0 = e0[4], l0 = e0[5];
  const float a1 = e1[0], b1 = e1[1], c1 = e1[2], h1 = e1[4], l1 = e1[5];
  const eu_i2 big = it >= 0x4c000000u;
  const eu_f2 num = { __builtin_fmaf(a0, t.x, b0), __builtin_fmaf(a1, t.y, b1) };
  float d0 = c0 * t.x, d1 = c1 * t.y;
  d0 = d0 + a0; d1 = d1 + a1;                 // d == a in every row of the table
  const eu_f2 den = { d0, d1 };
  eu_f2 x = eu_div2_safe(num, den);
  const float aT0 = 3.3333334327e-01f, aT1 = -2.0000000298e-01f, aT2 = 1.4285714924e-01f,
              aT3 = -1.1111110449e-01f, aT4 = 9.0908870101e-02f, aT5 = -7.6918758452e-02f,
              aT6 = 6.6610731184e-02f, aT7 = -5.8335702866e-02f, aT8 = 4.9768779427e-02f,
              aT9 = -3.6531571299e-02f, aT10 = 1.6285819933e-02f;
  eu_f2 z = x * x;
  eu_f2 w = z * z;
  eu_f2 s1 = z * (aT0 + w * (aT2 + w * (aT4 + w * (aT6 + w * (aT8 + w * aT10)))));
  eu_f2 s2 = w * (aT1 + w * (aT3 + w * (aT5 + w * (aT7 + w * aT9))));
  eu_f2 xs = x * (s1 + s2);
  eu_f2 u = { xs.x - l0, xs.y - l1 };
  u = u - x;
  const eu_f2 r = { h0 - u.x, h1 - u.y };
  const float hb = 1.5707962513e+00f + 7.5497894159e-08f;
  return eu_sel2(big, (eu_f2){ hb, hb }, r);
}

// atan2f with the table; x_positive: the caller guarantees x > 0 or out of range
EU_D2 eu_f2 eu_atan2f_2_tab(eu_f2 y, eu_f2 x, const float *tab, int x_positive)
{
  const eu_u2 hx = eu_bits2(x), hy = eu_bits2(y);
  const eu_u2 ix = hx & 0x7fffffffu, iy = hy & 0x7fffffffu;
  const eu_i2 okx = ((x_positive ? hx : ix) - 0x2b800000u) <= (0x53800000u - 0x2b800000u);
  const eu_i2 oky = (iy - 0x2b800000u) <= (0x53800000u - 0x2b800000u);
  const eu_i2 ok = okx & oky;
  // out-of-range lanes produce a value that is replaced below
  eu_f2 q = eu_div2_safe(y, x);
  eu_f2 z = eu_atanf_pos2_tab(eu_abs2(q), tab);
  eu_f2 r;
  if (x_positive) {
    r = eu_float2(eu_bits2(z) ^ (hy & 0x80000000u));
  } else {
    const float pi = 3.1415927410e+00f, pi_lo = -8.7422776573e-08f;
    const eu_i2 xneg = (eu_i2)(hx >> 31) != 0, yneg = (eu_i2)(hy >> 31) != 0;
    eu_f2 zl = z - pi_lo;
    eu_f2 rpos = eu_float2(eu_bits2(z) ^ (hy & 0x80000000u));
    eu_f2 rneg = eu_sel2(yneg, zl - pi, pi - zl);
    r = eu_sel2(xneg, rneg, rpos);
  }
  if (__builtin_expect(!(ok.x & ok.y), 0)) {
    if (!ok.x) r.x = eu_atan2f(y.x, x.x);
    if (!ok.y) r.y = eu_atan2f(y.y, x.y);
  }
  return r;
}


// ---------------------------------------------------------------------------
// Fast-path-only forms for kernels that hand the rare lanes to another kernel
// (eu_render4.hip): no scalar fallback, no branch; `ok` is cleared for lanes whose
// operands are outside the range the FMA sequences are exact for - the caller must
// not use those lanes' results.
// ---------------------------------------------------------------------------
EU_D2 eu_f2 eu_atan2f_2_tab_ok(eu_f2 y, eu_f2 x, const float *tab, int x_positive, eu_i2 &ok)
{
  const eu_u2 hx = eu_bits2(x), hy = eu_bits2(y);
  const eu_u2 ix = hx & 0x7fffffffu, iy = hy & 0x7fffffffu;
  const eu_i2 okx = ((x_positive ? hx : ix) - 0x2b800000u) <= (0x53800000u - 0x2b800000u);
  const eu_i2 oky = (iy - 0x2b800000u) <= (0x53800000u - 0x2b800000u);
  ok = ok & okx & oky;
  eu_f2 q = eu_div2_safe(y, x);
  eu_f2 z = eu_atanf_pos2_tab(eu_abs2(q), tab);
  if (x_positive) return eu_float2(eu_bits2(z) ^ (hy & 0x80000000u));
  const float pi = 3.1415927410e+00f, pi_lo = -8.7422776573e-08f;
  const eu_i2 xneg = (eu_i2)(hx >> 31) != 0, yneg = (eu_i2)(hy >> 31) != 0;
  eu_f2 zl = z - pi_lo;
  eu_f2 rpos = eu_float2(eu_bits2(z) ^ (hy & 0x80000000u));
  eu_f2 rneg = eu_sel2(yneg, zl - pi, pi - zl);
  return eu_sel2(xneg, rneg, rpos);
}

EU_D2 eu_f2 eu_div2_ok(eu_f2 n, eu_f2 d, eu_i2 &ok)
{
  const eu_u2 in = eu_bits2(n) & 0x7fffffffu, id = eu_bits2(d) & 0x7fffffffu;
  // d in [2^-40, 2^40]; n zero or in [2^-80, 2^40] (as eu_div2_guarded)
  const eu_i2 okd = (id - 0x2b800000u) <= (0x53800000u - 0x2b800000u);
  const eu_i2 okn = (in == 0u) | ((in - 0x17800000u) <= (0x53800000u - 0x17800000u));
  ok = ok & okd & okn;
  return eu_div2_safe(n, d);
}

EU_D2 eu_f2 eu_sqrt2_ok(eu_f2 x, eu_i2 &ok)
{
  const eu_u2 ix = eu_bits2(x);
  ok = ok & ((ix - 0x2b800000u) <= (0x53800000u - 0x2b800000u));
  return eu_sqrt2_safe(x);
}

// ---------------------------------------------------------------------------
// Round 3: the same functions with fewer vector instructions (eu_render5_kernel's FAST profile).
// Same operations on the same operands as eu_atanf_pos2_tab / eu_atan2f_2_tab_ok - only the
// instruction forms differ:
//   * the per-pixel table fields feed SCALAR operations (any register will do; as operands of
//     packed operations they first had to be moved into pairs: 12 v_mov per atanf),
//   * num = a * t + b with a in {0, 1, 2} is ONE fma (a * t is exact, so RN(a * t + b) is the
//     reference's RN(RN(a * t) + b)); den = RN(RN(c * t) + d) stays two operations (c = 1.5),
//   * the |t| < 7/16 row of the table has hi = lo = 0: hi - ((xs - lo) - x) = 0 - (xs - x) is
//     x - xs bit for bit, so no select,
//   * t >= 2^25 (the reference returns atanhi[3] + atanlo[3]) is reported in `big` instead of
//     being selected: the caller hands such a tile to the kernel with the fallbacks.
// ---------------------------------------------------------------------------
// A float literal formed where it is used (device: s_mov_b32 into a fresh scalar register, volatile so that it
// is neither hoisted out of the persistent loop nor kept live across it). The eleven coefficients of atanf's
// polynomial, kept live as loop invariants, are a quarter of the scalar registers eu_render5_kernel has, and
// what does not fit is spilled into VGPR lanes: 74 v_readlane / v_writelane per 16x16 tile.
#if defined(__HIP_DEVICE_COMPILE__)
#define EU_LIT(name, value) float name; asm volatile("s_mov_b32 %0, %1" : "=s"(name) : "i"(__builtin_bit_cast(int, (float)(value))))
#else
#define EU_LIT(name, value) const float name = (value)
#endif

// atanf(|q|) for both lanes; big.x / big.y: |q| >= 2^25 (the result of that lane is not valid)
EU_D2 eu_f2 eu_atanf_abs2_lean(eu_f2 q, const float *tab, eu_i2 &big)
{
  const eu_u2 iq = eu_bits2(q);
  const unsigned k0 = (iq.x >> 18) & 0x1fffu, k1 = (iq.y >> 18) & 0x1fffu;   // v_bfe_u32: sign excluded
  big = (eu_i2){ k0 >= 0x1300u ? -1 : 0, k1 >= 0x1300u ? -1 : 0 };           // 0x4c000000 >> 18
  const unsigned j0 = (k0 < 0xfb7u ? 0xfb7u : k0 > 0x1007u ? 0x1007u : k0) - 0xfb7u;
  const unsigned j1 = (k1 < 0xfb7u ? 0xfb7u : k1 > 0x1007u ? 0x1007u : k1) - 0xfb7u;
  const float *e0 = tab + j0 * 8, *e1 = tab + j1 * 8;
  const float a0 = e0[0], b0 = e0[1], c0 = e0[2], h0 = e0[4], l0 = e0[5];
  const float a1 = e1[0], b1 = e1[1], c1 = e1[2], h1 = e1[4], l1 = e1[5];
  const float t0 = __builtin_fabsf(q.x), t1 = __builtin_fabsf(q.y);
  const eu_f2 num = { __builtin_fmaf(a0, t0, b0), __builtin_fmaf(a1, t1, b1) };
  float d0 = c0 * t0, d1 = c1 * t1;
  d0 = d0 + a0; d1 = d1 + a1;                 // d == a in every row of the table
  const eu_f2 den = { d0, d1 };
  const eu_f2 x = eu_div2_safe(num, den);
  const eu_f2 z = x * x;
  const eu_f2 w = z * z;
  EU_LIT(aT0, 3.3333334327e-01f); EU_LIT(aT1, -2.0000000298e-01f); EU_LIT(aT2, 1.4285714924e-01f);
  EU_LIT(aT3, -1.1111110449e-01f); EU_LIT(aT4, 9.0908870101e-02f); EU_LIT(aT5, -7.6918758452e-02f);
  EU_LIT(aT6, 6.6610731184e-02f); EU_LIT(aT7, -5.8335702866e-02f); EU_LIT(aT8, 4.9768779427e-02f);
  EU_LIT(aT9, -3.6531571299e-02f); EU_LIT(aT10, 1.6285819933e-02f);
  const eu_f2 s1 = z * (aT0 + w * (aT2 + w * (aT4 + w * (aT6 + w * (aT8 + w * aT10)))));
  const eu_f2 s2 = w * (aT1 + w * (aT3 + w * (aT5 + w * (aT7 + w * aT9))));
  const eu_f2 xs = x * (s1 + s2);
  eu_f2 u = { xs.x - l0, xs.y - l1 };
  u = u - x;
  return (eu_f2){ h0 - u.x, h1 - u.y };
}

// atan2f(y, x), operands in range (the caller checks), x > 0 when x_positive
EU_D2 eu_f2 eu_atan2f_2_lean(eu_f2 y, eu_f2 x, const float *tab, int x_positive, eu_i2 &big)
{
  const eu_f2 q = eu_div2_safe(y, x);
  const eu_f2 z = eu_atanf_abs2_lean(q, tab, big);
  const eu_u2 hy = eu_bits2(y);
  eu_f2 m = z;
  if (!x_positive) {
    // m = 0 / 1: +-z; m = 2 / 3: +-(pi - (z - pi_lo))   (zl - pi = -(pi - zl) exactly)
    const float pi = 3.1415927410e+00f, pi_lo = -8.7422776573e-08f;
    const eu_f2 zl = z - pi_lo;
    const eu_f2 t = pi - zl;
    const eu_i2 xneg = (eu_i2)eu_bits2(x) < 0;
    m = eu_sel2(xneg, t, z);
  }
  // z >= +0 and t > 0: the result is m with y's sign bit
  return eu_float2((eu_bits2(m) & 0x7fffffffu) | (hy & 0x80000000u));
}

// x / c with r = the refined reciprocal of c as LLVM's division computes it (eu_rcp_refined):
// the rest of eu_div2_safe's sequence, i.e. the correctly rounded quotient for x in {0} u
// [2^-90, 2^90] and c in [2^-90, 2^90]
EU_D2 eu_f2 eu_div2_rr(eu_f2 x, float c, float r)
{
  const eu_f2 cc = { c, c }, rr = { r, r };
  eu_f2 q = x * rr;
  eu_f2 e = eu_fma2(-cc, q, x);
  q = eu_fma2(e, rr, q);
  e = eu_fma2(-cc, q, x);
  return eu_fma2(e, rr, q);
}

#if defined(__HIPCC__)
// the reciprocal eu_div2_safe forms from v_rcp_f32 before it multiplies (device only)
__device__ __forceinline__ float eu_rcp_refined(float d)
{
  float r = __builtin_amdgcn_rcpf(d);
  const float e = __builtin_fmaf(-d, r, 1.0f);
  return __builtin_fmaf(e, r, r);
}

// n / d: range-checked FMA division (eu_div2_safe), hipcc's correctly rounded
// `/` for the lanes outside the range - the same bits either way
__device__ __forceinline__ eu_f2 eu_div2_guarded(eu_f2 n, eu_f2 d)
{
  const eu_u2 in = eu_bits2(n) & 0x7fffffffu, id = eu_bits2(d) & 0x7fffffffu;
  // d in [2^-40, 2^40]; n zero or in [2^-80, 2^40]
  const eu_i2 okd = (id - 0x2b800000u) <= (0x53800000u - 0x2b800000u);
  const eu_i2 okn = (in == 0u) | ((in - 0x17800000u) <= (0x53800000u - 0x17800000u));
  const eu_i2 ok = okd & okn;
  const eu_f2 one = { 1.0f, 1.0f };
  eu_f2 q = eu_div2_safe(eu_sel2(ok, n, one), eu_sel2(ok, d, one));
  if (__builtin_expect(!(ok.x & ok.y), 0)) {
    if (!ok.x) q.x = n.x / d.x;
    if (!ok.y) q.y = n.y / d.y;
  }
  return q;
}

// sqrt(x): eu_sqrt2_safe for x in [2^-40, 2^40], sqrtf for the other lanes
__device__ __forceinline__ eu_f2 eu_sqrt2_guarded(eu_f2 x)
{
  const eu_u2 ix = eu_bits2(x);
  const eu_i2 ok = (ix - 0x2b800000u) <= (0x53800000u - 0x2b800000u);
  const eu_f2 one = { 1.0f, 1.0f };
  eu_f2 r = eu_sqrt2_safe(eu_sel2(ok, x, one));
  if (__builtin_expect(!(ok.x & ok.y), 0)) {
    if (!ok.x) r.x = sqrtf(x.x);
    if (!ok.y) r.y = sqrtf(x.y);
  }
  return r;
}
#endif

#endif

// Device functions of the packed (two pixels per lane) render kernels: weights,
// LDS / global texel reads, the weighted sum, gates, rays from the stepper tables and
// the ray -> source coordinate stage on float2 operands. Shared by eu_render2.hip
// (direct gathers) and eu_render4.hip (per-wave LDS staging).
#ifndef EU_PACKED_DEV_H
#define EU_PACKED_DEV_H

#ifndef EU_CUBEFACE_BALLOT
#define EU_CUBEFACE_BALLOT 0   // eu_coord2_ok: wavefront ballot on the cube-face dominance class (measured: DESIGN.md 5)
#endif

#include "eu_render_dev.h"
#include "eu_math2.h"

// EU_FMA_EXPERIMENT (a labelled experiment, never the shipped build): the b-spline weights and the weighted sum
// with fused multiply-adds - what north_star's tolerance (1 ULP per channel for the interpolation) would
// allow, against the 0-ULP contract the library keeps. Offsets, deltas and everything in front of
// them are untouched. EU_MAD(a, b, c) = a * b + c in the contract's two roundings, or fused.
#ifdef EU_FMA_EXPERIMENT
#define EU_MAD2(a, b, c) __builtin_elementwise_fma((a), (b), (c))
#define EU_MAD1(a, b, c) __builtin_fmaf((a), (b), (c))
#else
#define EU_MAD2(a, b, c) ((a) * (b) + (c))
#define EU_MAD1(a, b, c) ((a) * (b) + (c))
#endif

typedef const __attribute__((address_space(4))) float *eu_cptr;   // scalar-cache loads

// weights of both lanes for one axis; DEG 2 and 3 use the literal weight
// matrix (zimt/basis.h:419-545 evaluated in long double, narrowed to float;
// tests/test_abi.py compares the literals with eu::weight_matrix)
template <int DEG>
__device__ __forceinline__ void eu_weights2(const float *wm, eu_f2 d, eu_f2 *w)
{
  if constexpr (DEG == 3) {
    const float a = 0x1.555556p-3f, b = 0x1.555556p-1f;
    eu_f2 d2 = d * d, d3 = d2 * d;
    const eu_f2 mh = { -0.5f, -0.5f }, ph = { 0.5f, 0.5f }, ma = { -a, -a }, pa = { a, a };
    eu_f2 w0 = EU_MAD2(d, mh, pa); w0 = EU_MAD2(d2, ph, w0); w0 = EU_MAD2(d3, ma, w0);
    eu_f2 w1 = b - d2;             w1 = EU_MAD2(d3, ph, w1);
    eu_f2 w2 = EU_MAD2(d, ph, pa); w2 = EU_MAD2(d2, ph, w2); w2 = EU_MAD2(d3, mh, w2);
    w[0] = w0; w[1] = w1; w[2] = w2; w[3] = d3 * a;
  } else if constexpr (DEG == 2) {
    eu_f2 d2 = d * d;
    eu_f2 w0 = 0.125f + d * -0.5f; w0 = w0 + d2 * 0.5f;
    eu_f2 w1 = 0.75f - d2;
    eu_f2 w2 = 0.125f + d * 0.5f;  w2 = w2 + d2 * 0.5f;
    w[0] = w0; w[1] = w1; w[2] = w2;
  } else {
    constexpr int order = DEG + 1;
#pragma unroll
    for (int c = 0; c <= DEG; c++) w[c] = (eu_f2){ wm[c * order], wm[c * order] };
    eu_f2 power = d;
#pragma unroll
    for (int row = 1; row <= DEG; row++) {
#pragma unroll
      for (int c = 0; c <= DEG; c++) w[c] = w[c] + power * wm[c * order + row];
      if (row < DEG) power = power * d;
    }
  }
}

// LDS texel read: one aligned 16-byte ds_read_b128 for RGB(X) / RGBA texels
template <int NCH, int TS, class PTR>
__device__ __forceinline__ void eu_texel(PTR q, float *t)
{
  if constexpr (TS == 4 && sizeof(PTR) == 4) {
    typedef float eu_f4 __attribute__((ext_vector_type(4)));
    typedef const __attribute__((address_space(3))) eu_f4 *eu_l4ptr;
    eu_f4 v = *(eu_l4ptr)q;
    t[0] = v.x; t[1] = v.y; t[2] = v.z;
    if constexpr (NCH == 4) t[3] = v.w;
  } else {
#pragma unroll
    for (int c = 0; c < NCH; c++) t[c] = q[c];
  }
}

// weighted sum for one pixel: channels 0/1 packed, the rest scalar
typedef const __attribute__((address_space(3))) float *eu_lptr;   // LDS address space: ds_read, not flat

template <int NCH, int DEG, int TS = NCH, class STRIDE = long long, class PTR = const float *>
__device__ __forceinline__ void eu_accumulate1(PTR p0, STRIDE es1,
                                               const float *wx, const float *wy, float tx,
                                               float ty, float *out)
{
  if constexpr (DEG == 1) {
    float wl0 = 1.0f - tx, wr0 = tx, wl1 = 1.0f - ty, wr1 = ty;
    PTR q = p0 + es1;
    float a[NCH], b[NCH], c2[NCH], d[NCH];
    eu_texel<NCH, TS, PTR>(p0, a); eu_texel<NCH, TS, PTR>(p0 + TS, b);
    eu_texel<NCH, TS, PTR>(q, c2); eu_texel<NCH, TS, PTR>(q + TS, d);
#pragma unroll
    for (int c = 0; c < NCH; c++) {
      float sum = a[c] * wl0;
      sum = EU_MAD1(b[c], wr0, sum);
      sum = sum * wl1;
      float sub = c2[c] * wl0;
      sub = EU_MAD1(d[c], wr0, sub);
      sum = EU_MAD1(sub, wr1, sum);
      out[c] = sum;
    }
  } else {
    constexpr int order = DEG + 1;
    float sum[NCH];
#pragma unroll
    for (int j = 0; j < order; j++) {
      PTR rowp = p0 + j * es1;
      float t[order][NCH];
#pragma unroll
      for (int i = 0; i < order; i++) eu_texel<NCH, TS, PTR>(rowp + i * TS, t[i]);
#pragma unroll
      for (int c = 0; c < NCH; c++) {
        float r = t[0][c] * wx[0];
#pragma unroll
        for (int i = 1; i < order; i++) r = EU_MAD1(wx[i], t[i][c], r);
        if (j == 0) sum[c] = r * wy[0];
        else sum[c] = EU_MAD1(r, wy[j], sum[c]);
      }
    }
#pragma unroll
    for (int c = 0; c < NCH; c++) out[c] = sum[c];
  }
}

__device__ __forceinline__ float eu_gate1(float c, int kind, float lower, float upper)
{
  return eu_gate(c, kind, lower, upper);
}

// gate for both lanes: the common case (inside [lower, upper)) is two packed
// operations; lanes that need folding take the scalar gate
__device__ __forceinline__ eu_f2 eu_gate2(eu_f2 c, int kind, float lower, float upper)
{
  if (kind == 0) {
    eu_f2 r = c;
    r.x = eu_gate1(c.x, 0, lower, upper);
    r.y = eu_gate1(c.y, 0, lower, upper);
    return r;
  }
  const float w = upper - lower;
  eu_f2 cc = c - lower;
  if (kind == 1) cc = eu_abs2(cc);
  eu_i2 out = kind == 2 ? ((cc < 0.0f) | (cc >= w)) : (cc >= w);
  eu_f2 r = cc + lower;
  if (__builtin_expect(out.x | out.y, 0)) {
    if (out.x) r.x = eu_gate1(c.x, kind, lower, upper);
    if (out.y) r.y = eu_gate1(c.y, kind, lower, upper);
  }
  return r;
}

// ---------------------------------------------------------------------------
// both lanes: rays from the stepper tables
// ---------------------------------------------------------------------------

struct eu_ray2 { eu_f2 x, y, z; };

// rowt: A[3], B[3], C[3] of one stepper (scalar-cache loads, wave-uniform)
__device__ __forceinline__ eu_ray2 eu_rays2(int form, int norm_mode, eu_cptr rowt,
                                            const float *__restrict__ colA,
                                            const float *__restrict__ colB, int xa, int xb)
{
  const float A0 = rowt[0], A1 = rowt[1], A2 = rowt[2], B0 = rowt[3], B1 = rowt[4], B2 = rowt[5];
  const eu_f2 c0 = { colA[xa], colA[xb] };
  eu_ray2 r;
  float C0 = 0.0f, C1 = 0.0f, C2 = 0.0f;
  eu_f2 c1 = { 0.0f, 0.0f };
  if (form == EU_FORM_BCA) {
    C0 = rowt[6]; C1 = rowt[7]; C2 = rowt[8];
    c1 = (eu_f2){ colB[xa], colB[xb] };
    r.x = B0 * c0 + C0 * c1 + A0;
    r.y = B1 * c0 + C1 * c1 + A1;
    r.z = B2 * c0 + C2 * c1 + A2;
  } else {
    r.x = B0 * c0 + A0;
    r.y = B1 * c0 + A1;
    r.z = B2 * c0 + A2;
  }
  if (norm_mode == EU_NORM_DIV) {
    // trg /= norm(trg), xel.h:752-765 (rectilinear / cubemap / biatan6 steppers
    // with normalize = true)
    eu_f2 sqn = r.x * r.x; sqn = sqn + r.y * r.y; sqn = sqn + r.z * r.z;
    eu_f2 n = { sqrtf(sqn.x), sqrtf(sqn.y) };
    r.x = r.x / n; r.y = r.y / n; r.z = r.z / n;
  } else if (norm_mode == EU_NORM_CYL) {
    // cylindrical_stepper: reciprocal length of the lane's FIRST pixel in the
    // 512-pixel segment (stepper.h:771-775, :786)
    int sa = (xa / EU_SEGMENT) * EU_SEGMENT, sb = (xb / EU_SEGMENT) * EU_SEGMENT;
    int fa = sa + ((xa - sa) % EU_LANES), fb = sb + ((xb - sb) % EU_LANES);
    const eu_f2 d0 = { colA[fa], colA[fb] }, d1 = { colB[fa], colB[fb] };
    eu_f2 fx = B0 * d0 + C0 * d1 + A0, fy = B1 * d0 + C1 * d1 + A1, fz = B2 * d0 + C2 * d1 + A2;
    eu_f2 sqn = fx * fx; sqn = sqn + fy * fy; sqn = sqn + fz * fz;
    eu_f2 rcp = { 1.0f / sqrtf(sqn.x), 1.0f / sqrtf(sqn.y) };
    r.x = r.x * rcp; r.y = r.y * rcp; r.z = r.z * rcp;
  }
  return r;
}

// x / c for an extent c: eu_div2_rr where its operands are in range (x zero or in [2^-40, 2^40], c in
// [2^-20, 2^20]: then the same bits as `/`), `/` for the other lanes
__device__ __forceinline__ eu_f2 eu_div2_ext(eu_f2 x, float c)
{
  const eu_u2 ix = eu_bits2(x) & 0x7fffffffu;
  const eu_i2 okx = (ix == 0u) | ((ix - 0x2b800000u) <= (0x53800000u - 0x2b800000u));
  const bool okc = c >= 0x1p-20f && c <= 0x1p20f;
  eu_f2 q = eu_div2_rr(x, c, eu_rcp_refined(c));
  if (__builtin_expect(!(okc && okx.x && okx.y), 0)) {
    if (!(okc && okx.x)) q.x = x.x / c;
    if (!(okc && okx.y)) q.y = x.y / c;
  }
  return q;
}

// ---------------------------------------------------------------------------
// both lanes: ray -> source pixel coordinate (+ hit mask)
// ---------------------------------------------------------------------------

template <int PRJ>
__device__ __forceinline__ eu_i2 eu_coord2(const eu_src_dev &s, const eu_ray2 &r, eu_f2 &sx,
                                           eu_f2 &sy, const float *atab)
{
  if constexpr (PRJ == EU_CUBEMAP || PRJ == EU_BIATAN6) {
    // ray_to_cubeface, geometry.h:1178-1289 (dominance classes by select)
    const eu_f2 ax = eu_abs2(r.x), ay = eu_abs2(r.y), az = eu_abs2(r.z);
    const eu_i2 m1 = ax >= ay, m2 = ax >= az, m3 = ay >= az;
    const eu_i2 domx = m1 & m2, domz = (~m2) & (~m3);
    const eu_f2 num0 = eu_sel2(domx, -r.z, eu_sel2(domz, r.x, -r.x));
    const eu_f2 den0 = eu_sel2(domx, r.x, eu_sel2(domz, r.z, ay));
    const eu_f2 num1 = eu_sel2(domx, r.y, eu_sel2(domz, r.y, r.z));
    const eu_f2 den1 = eu_sel2(domx, ax, eu_sel2(domz, az, r.y));
    eu_f2 in0 = eu_div2_guarded(num0, den0), in1 = eu_div2_guarded(num1, den1);
    const eu_i2 fx = eu_sel2i(r.x < 0.0f, 0, 1), fz = eu_sel2i(r.z < 0.0f, 5, 4),
                fy = eu_sel2i(r.y < 0.0f, 2, 3);
    const eu_i2 face = eu_sel2i(domx, fx, eu_sel2i(domz, fz, fy));
    if constexpr (PRJ == EU_BIATAN6) {
      // in_face = float(4/pi) * atan(in_face), environment.h:1480; atanf is odd
      const float k = (float)(4.0 / 3.14159265358979323846);
      eu_f2 a0 = eu_atanf_pos2_tab(eu_abs2(in0), atab), a1 = eu_atanf_pos2_tab(eu_abs2(in1), atab);
      a0 = eu_float2(eu_bits2(a0) | (eu_bits2(in0) & 0x80000000u));
      a1 = eu_float2(eu_bits2(a1) | (eu_bits2(in1) & 0x80000000u));
      in0 = k * a0; in1 = k * a1;
    }
    // cubemap_view_t::get_pickup_coordinate_px, environment.h:1452-1460
    eu_f2 p0 = in0 + s.refc_md, p1 = in1 + s.refc_md;
    p0 = p0 * s.model_to_px; p1 = p1 * s.model_to_px;
    const eu_i2 fs = face * s.section_px;
    p1 = p1 + (eu_f2){ (float)fs.x, (float)fs.y };
    sx = p0 - .5f; sy = p1 - .5f;
    return (eu_i2){ -1, -1 };
  } else {
    // ray_to_ll_t (geometry.h:278-301): s = sqrt(r*r + f*f); lat = atan2(d, s); lon = atan2(r, f)
    eu_f2 q2 = r.x * r.x + r.z * r.z;
    const eu_f2 qs = eu_sqrt2_guarded(q2);
    // s == 0 (ray along the vertical axis) fails the range check and takes the scalar path
    eu_f2 lat = eu_atan2f_2_tab(r.y, qs, atab, 1);
    eu_f2 lon = eu_atan2f_2_tab(r.x, r.z, atab, 0);
    // a full-sphere image covers atan2f's whole range: every ray hits
    eu_i2 hit = { -1, -1 };
    if (!s.always_hit)
      hit = (lon >= s.wex0) & (lon <= s.wex1) & (lat >= s.wex2) & (lat <= s.wex3);
    // source_t::md_to_spline (environment.h:988-1006)
    eu_f2 i0 = { (float)((double)lon.x - s.tex_x0), (float)((double)lon.y - s.tex_x0) };
    eu_f2 i1 = { (float)((double)lat.x - s.tex_y0), (float)((double)lat.y - s.tex_y0) };
    if (s.cdiv_ok) {
      i0 = eu_div2_const(i0, s.ext_w, s.rcp_ext_w);
      i1 = eu_div2_const(i1, s.ext_h, s.rcp_ext_h);
    } else {
      // the constant division is not exact for this extent (2 pi and pi are such: every full-sphere
      // source): the FMA division with the reciprocal formed once, hipcc's `/` for a lane out of range
      i0 = eu_div2_ext(i0, s.ext_w);
      i1 = eu_div2_ext(i1, s.ext_h);
    }
    i0 = i0 * s.total_w; i0 = i0 - .5f;
    i1 = i1 * s.total_h; i1 = i1 - .5f;
    sx = i0 - s.win_x_off; sy = i1 - s.win_y_off;
    return hit;
  }
}

// ---------------------------------------------------------------------------
// both lanes: b-spline evaluation at (sx, sy); misses give 0
// ---------------------------------------------------------------------------

template <int NCH, int DEG>
__device__ __forceinline__ void eu_eval2(const eu_src_dev &s, eu_f2 sx, eu_f2 sy, eu_i2 hit,
                                         float *pxa, float *pxb)
{
  // gate + split (map.h, basis.h:102-146)
  eu_f2 gx = eu_gate2(sx, s.gate0, s.lower0, s.upper0);
  eu_f2 gy = eu_gate2(sy, s.gate1, s.lower1, s.upper1);
  eu_f2 fx, fy;
  if constexpr (DEG & 1) {
    fx = (eu_f2){ floorf(gx.x), floorf(gx.y) }; fy = (eu_f2){ floorf(gy.x), floorf(gy.y) };
  } else {
    fx = (eu_f2){ roundf(gx.x), roundf(gx.y) }; fy = (eu_f2){ roundf(gy.x), roundf(gy.y) };
  }
  const eu_f2 tx = gx - fx, ty = gy - fy;
  constexpr int order = DEG + 1;
  eu_f2 wx[order], wy[order];
  if constexpr (DEG >= 2) {
    eu_weights2<DEG>(s.wm, tx, wx);
    eu_weights2<DEG>(s.wm, ty, wy);
  }
  float wxa[order], wxb[order], wya[order], wyb[order];
#pragma unroll
  for (int i = 0; i < order; i++) {
    if constexpr (DEG >= 2) { wxa[i] = wx[i].x; wxb[i] = wx[i].y; wya[i] = wy[i].x; wyb[i] = wy[i].y; }
    else { wxa[i] = wxb[i] = wya[i] = wyb[i] = 0.0f; }
  }
  // a missed lane's coordinate is arbitrary (the reference evaluates it at an
  // uninitialised but gated position and zeroes the result)
  // a lane without a hit reads the window at the core origin (always inside
  // the container, framed or not); its result is discarded
  const int ixa = hit.x ? (int)fx.x : DEG / 2, iya = hit.x ? (int)fy.x : DEG / 2;
  const int ixb = hit.y ? (int)fx.y : DEG / 2, iyb = hit.y ? (int)fy.y : DEG / 2;
  const float *pa = s.base + (long long)(ixa - DEG / 2) * NCH + (long long)(iya - DEG / 2) * s.es1;
  const float *pb = s.base + (long long)(ixb - DEG / 2) * NCH + (long long)(iyb - DEG / 2) * s.es1;
  eu_accumulate1<NCH, DEG>(pa, s.es1, wxa, wya, tx.x, ty.x, pxa);
  eu_accumulate1<NCH, DEG>(pb, s.es1, wxb, wyb, tx.y, ty.y, pxb);
  // environment::eval brighten (environment.h:1821-1842), zero on a miss
  constexpr int ncol = (NCH == 2 || NCH == 4) ? NCH - 1 : NCH;
  const bool bright = s.brighten != 1.0f;
#pragma unroll
  for (int c = 0; c < NCH; c++) {
    float va = pxa[c], vb = pxb[c];
    if (bright && c < ncol) { va = va * s.brighten; vb = vb * s.brighten; }
    pxa[c] = hit.x ? va : 0.0f;
    pxb[c] = hit.y ? vb : 0.0f;
  }
}


// ---------------------------------------------------------------------------
// The same two stages without any scalar fallback (eu_render4.hip, staged kernel): lanes
// that would need one clear `ok`; their tile is rendered by the kernel that has the
// fallbacks. Bit-identical to eu_gate2 / eu_coord2 on the lanes that stay ok.
// ---------------------------------------------------------------------------
__device__ __forceinline__ eu_f2 eu_gate2_ok(eu_f2 c, int kind, float lower, float upper, eu_i2 &ok)
{
  if (kind == 0) {            // clamp_gate, map.h:231-236
    eu_f2 r = c;
    r = eu_sel2(c < lower, (eu_f2){ lower, lower }, r);
    r = eu_sel2(c > upper, (eu_f2){ upper, upper }, r);
    return r;
  }
  const float w = upper - lower;
  eu_f2 cc = c - lower;
  if (kind == 1) cc = eu_abs2(cc);
  const eu_i2 out = kind == 2 ? ((cc < 0.0f) | (cc >= w)) : (cc >= w);
  ok = ok & ~out;
  return cc + lower;
}

template <int PRJ, bool FAST = false>
__device__ __forceinline__ eu_i2 eu_coord2_ok(const eu_src_dev &s, const eu_ray2 &r, eu_f2 &sx,
                                              eu_f2 &sy, const float *atab, eu_i2 &ok)
{
  if constexpr (PRJ == EU_CUBEMAP || PRJ == EU_BIATAN6) {
    const eu_f2 ax = eu_abs2(r.x), ay = eu_abs2(r.y), az = eu_abs2(r.z);
    const eu_i2 m1 = ax >= ay, m2 = ax >= az, m3 = ay >= az;
    const eu_i2 domx = m1 & m2, domz = (~m2) & (~m3);
    eu_f2 num0, den0, num1, den1;
    eu_i2 face;
#if EU_CUBEFACE_BALLOT
    // wavefront ballot on the dominance class (the GPU form of the reference's any_of(dom)
    // early-outs, geometry.h:1224-1287): a wave whose 128 pixels agree - all but the waves on a
    // cube edge - takes that class's operands without the select ladder
    const unsigned long long bx = __ballot(domx.x && domx.y), bz = __ballot(domz.x && domz.y);
    const unsigned long long by = __ballot(!(domx.x | domx.y | domz.x | domz.y)), all = __ballot(1);
    if (bx == all) {
      num0 = -r.z; den0 = r.x; num1 = r.y; den1 = ax; face = eu_sel2i(r.x < 0.0f, 0, 1);
    } else if (bz == all) {
      num0 = r.x; den0 = r.z; num1 = r.y; den1 = az; face = eu_sel2i(r.z < 0.0f, 5, 4);
    } else if (by == all) {
      num0 = -r.x; den0 = ay; num1 = r.z; den1 = r.y; face = eu_sel2i(r.y < 0.0f, 2, 3);
    } else
#endif
    {
      num0 = eu_sel2(domx, -r.z, eu_sel2(domz, r.x, -r.x));
      den0 = eu_sel2(domx, r.x, eu_sel2(domz, r.z, ay));
      num1 = eu_sel2(domx, r.y, eu_sel2(domz, r.y, r.z));
      den1 = eu_sel2(domx, ax, eu_sel2(domz, az, r.y));
      const eu_i2 fx = eu_sel2i(r.x < 0.0f, 0, 1), fz = eu_sel2i(r.z < 0.0f, 5, 4),
                  fy = eu_sel2i(r.y < 0.0f, 2, 3);
      face = eu_sel2i(domx, fx, eu_sel2i(domz, fz, fy));
    }
    eu_f2 in0 = eu_div2_ok(num0, den0, ok), in1 = eu_div2_ok(num1, den1, ok);
    if constexpr (PRJ == EU_BIATAN6) {
      const float k = (float)(4.0 / 3.14159265358979323846);
      eu_f2 a0 = eu_atanf_pos2_tab(eu_abs2(in0), atab), a1 = eu_atanf_pos2_tab(eu_abs2(in1), atab);
      a0 = eu_float2(eu_bits2(a0) | (eu_bits2(in0) & 0x80000000u));
      a1 = eu_float2(eu_bits2(a1) | (eu_bits2(in1) & 0x80000000u));
      in0 = k * a0; in1 = k * a1;
    }
    eu_f2 p0 = in0 + s.refc_md, p1 = in1 + s.refc_md;
    p0 = p0 * s.model_to_px; p1 = p1 * s.model_to_px;
    const eu_i2 fs = face * s.section_px;
    p1 = p1 + (eu_f2){ (float)fs.x, (float)fs.y };
    sx = p0 - .5f; sy = p1 - .5f;
    return (eu_i2){ -1, -1 };
  } else {
    eu_f2 q2 = r.x * r.x + r.z * r.z;
    const eu_f2 qs = eu_sqrt2_ok(q2, ok);
    eu_f2 lat = eu_atan2f_2_tab_ok(r.y, qs, atab, 1, ok);
    eu_f2 lon = eu_atan2f_2_tab_ok(r.x, r.z, atab, 0, ok);
    eu_i2 hit = { -1, -1 };
    if (!FAST && !s.always_hit)
      hit = (lon >= s.wex0) & (lon <= s.wex1) & (lat >= s.wex2) & (lat <= s.wex3);
    eu_f2 i0 = { (float)((double)lon.x - s.tex_x0), (float)((double)lon.y - s.tex_x0) };
    eu_f2 i1 = { (float)((double)lat.x - s.tex_y0), (float)((double)lat.y - s.tex_y0) };
    if (FAST || s.cdiv_ok) {
      i0 = eu_div2_const(i0, s.ext_w, s.rcp_ext_w);
      i1 = eu_div2_const(i1, s.ext_h, s.rcp_ext_h);
    } else {
      // (as eu_div2_ext, with the range test reported in `ok` instead of a fallback)
      const eu_u2 j0 = eu_bits2(i0) & 0x7fffffffu, j1 = eu_bits2(i1) & 0x7fffffffu;
      const eu_i2 in0 = (j0 == 0u) | ((j0 - 0x2b800000u) <= (0x53800000u - 0x2b800000u));
      const eu_i2 in1 = (j1 == 0u) | ((j1 - 0x2b800000u) <= (0x53800000u - 0x2b800000u));
      const int okc = (s.ext_w >= 0x1p-20f && s.ext_w <= 0x1p20f && s.ext_h >= 0x1p-20f && s.ext_h <= 0x1p20f) ? -1 : 0;
      ok = ok & in0 & in1 & (eu_i2){ okc, okc };
      i0 = eu_div2_rr(i0, s.ext_w, eu_rcp_refined(s.ext_w));
      i1 = eu_div2_rr(i1, s.ext_h, eu_rcp_refined(s.ext_h));
    }
    i0 = i0 * s.total_w; i0 = i0 - .5f;
    i1 = i1 * s.total_h; i1 = i1 - .5f;
    sx = i0 - s.win_x_off; sy = i1 - s.win_y_off;
    return hit;
  }
}

#endif

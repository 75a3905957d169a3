// Device functions of the render path (included by eu_render.hip and the
// diagnostic kernels in eu_diag.hip). See eu_render.hip for the design notes.
#ifndef EU_RENDER_DEV_H
#define EU_RENDER_DEV_H

#include <hip/hip_runtime.h>
#include <climits>
#include "eu_device.h"
#include "eu_math.h"
#include "eu_math2.h"

#ifndef EU_COORD_LEAN
#define EU_COORD_LEAN 0      // 1: eu_source_coordinate with range-checked FMA forms of its divisions and square roots (measured: config 5 7.3 -> 9.3 ms with EU_COORD_SINCOS, the multi-facet kernel spills three times as much)
#endif
#ifndef EU_COORD_SINCOS
#define EU_COORD_SINCOS 0    // 1: the fisheye mounts sinf and cosf of one angle from one reduction (config 5: 7.3 -> 8.7 ms)
#endif

#define EU_TILE_W 64
#define EU_TILE_H 4

// EU_FMA_EXPERIMENT (a labelled experiment, never the shipped build; eu_packed_dev.h): the weighted sum fused
#ifdef EU_FMA_EXPERIMENT
#define EU_MADS(a, b, c) __builtin_fmaf((a), (b), (c))
#else
#define EU_MADS(a, b, c) ((a) * (b) + (c))
#endif

// ---------------------------------------------------------------------------
// gates: zimt/map.h:184-440
// ---------------------------------------------------------------------------

__device__ __forceinline__ float eu_vfmod(float lhs, float rhs)
{
  float help = lhs / rhs;
  help = truncf(help);
  help = help * rhs;
  lhs = lhs - help;
  if (fabsf(lhs) >= fabsf(rhs)) lhs = 0.0f;
  return lhs;
}

__device__ __forceinline__ float eu_gate(float c, int kind, float lower, float upper)
{
  if (kind == 2) {            // periodic_gate, map.h:423-440
    float cc = c - lower;
    float w = upper - lower;
    bool below = cc < 0.0f, above = cc >= w;
    if (below || above) {
      float cm = eu_vfmod(cc, w);
      if (below) cm = cm + w;
      if (cm >= w) cm = 0.0f;
      cc = cm;
    }
    return cc + lower;
  }
  if (kind == 1) {            // mirror_gate, map.h:341-357
    float cc = c - lower;
    float w = upper - lower;
    cc = fabsf(cc);
    if (cc >= w) {
      float cm = eu_vfmod(cc, 2 * w);
      cm = cm - w;
      cm = fabsf(cm);
      cm = w - cm;
      cc = cm;
    }
    return cc + lower;
  }
  float r = c;                // clamp_gate, map.h:231-236
  if (c < lower) r = lower;
  if (c > upper) r = upper;
  return r;
}

// ---------------------------------------------------------------------------
// b-spline evaluation: split (basis.h:102-146), weights (basis.h:650-690),
// offsets and weighted sum (eval.h:904-1059, :1237-1300)
// ---------------------------------------------------------------------------

template <int DEG>
__device__ __forceinline__ void eu_weights(const float *wm, float delta, float *w)
{
  constexpr int order = DEG + 1;
#pragma unroll
  for (int c = 0; c <= DEG; c++) w[c] = wm[c * order];
  float power = delta;
#pragma unroll
  for (int row = 1; row <= DEG; row++) {
#pragma unroll
    for (int c = 0; c <= DEG; c++) w[c] = w[c] + power * wm[c * order + row];
    if (row < DEG) power = power * delta;
  }
}

// gate + split: coordinate -> integer base position and fractional parts
template <int DEG>
__device__ __forceinline__ void eu_split(const eu_src_dev &s, float cx, float cy, int &ix,
                                         int &iy, float &tx, float &ty)
{
  float gx = eu_gate(cx, s.gate0, s.lower0, s.upper0);
  float gy = eu_gate(cy, s.gate1, s.lower1, s.upper1);
  float fx = (DEG & 1) ? floorf(gx) : roundf(gx);
  float fy = (DEG & 1) ? floorf(gy) : roundf(gy);
  tx = gx - fx; ty = gy - fy;
  ix = (int)fx; iy = (int)fy;
}

// tap (i, j) of the (DEG+1)^2 window straight from the braced container
template <int NCH>
struct eu_global_taps {
  const float *p0; long long es0, es1;
  __device__ __forceinline__ void load(int j, int i, float *t) const {
    const float *q = p0 + j * es1 + i * es0;
#pragma unroll
    for (int c = 0; c < NCH; c++) t[c] = q[c];
  }
};

// ... or from the workgroup's LDS copy of the source bounding box; texels are
// padded to TEX floats so that RGB reads are one aligned ds_read_b128
template <int NCH, int TEX>
struct eu_lds_taps {
  const float *p0; int pitch;       // floats
  __device__ __forceinline__ void load(int j, int i, float *t) const {
    const float *q = p0 + j * pitch + i * TEX;
    if constexpr (TEX == 4) {
      float4 v = *reinterpret_cast<const float4 *>(q);
      t[0] = v.x; t[1] = v.y; t[2] = v.z;
      if constexpr (NCH == 4) t[3] = v.w;
    } else if constexpr (TEX == 2) {
      float2 v = *reinterpret_cast<const float2 *>(q);
      t[0] = v.x; t[1] = v.y;
    } else {
      t[0] = q[0];
    }
  }
};

// weighted sum over the window, in the reference's order (eval.h:904-1059)
template <int NCH, int DEG, class TAPS>
__device__ __forceinline__ void eu_accumulate(const float *wm, float tx, float ty,
                                              const TAPS &taps, float *out)
{
  if constexpr (DEG == 0) {
    taps.load(0, 0, out);
  } else if constexpr (DEG == 1) {
    float wl0 = 1.0f - tx, wr0 = tx, wl1 = 1.0f - ty, wr1 = ty;
    float a[NCH], b[NCH], c2[NCH], d[NCH];
    taps.load(0, 0, a); taps.load(0, 1, b); taps.load(1, 0, c2); taps.load(1, 1, d);
#pragma unroll
    for (int c = 0; c < NCH; c++) {
      float sum = a[c] * wl0;
      sum = EU_MADS(b[c], wr0, sum);
      sum = sum * wl1;
      float sub = c2[c] * wl0;
      sub = EU_MADS(d[c], wr0, sub);
      sum = EU_MADS(sub, wr1, sum);
      out[c] = sum;
    }
  } else {
    constexpr int order = DEG + 1;
    float wx[order], wy[order];
    eu_weights<DEG>(wm, tx, wx);
    eu_weights<DEG>(wm, ty, wy);
    float sum[NCH];
#pragma unroll
    for (int j = 0; j < order; j++) {
      float t[order][NCH];
#pragma unroll
      for (int i = 0; i < order; i++) taps.load(j, i, t[i]);
#pragma unroll
      for (int c = 0; c < NCH; c++) {
        float r = t[0][c] * wx[0];
#pragma unroll
        for (int i = 1; i < order; i++) r = EU_MADS(wx[i], t[i][c], r);
        if (j == 0) sum[c] = r * wy[0];
        else sum[c] = EU_MADS(r, wy[j], sum[c]);
      }
    }
#pragma unroll
    for (int c = 0; c < NCH; c++) out[c] = sum[c];
  }
}

template <int NCH, int DEG>
__device__ __forceinline__ void eu_bspline(const eu_src_dev &s, float cx, float cy,
                                           float *out)
{
  int ix, iy;
  float tx, ty;
  eu_split<DEG>(s, cx, cy, ix, iy, tx, ty);
  eu_global_taps<NCH> g;
  g.es0 = s.es0; g.es1 = s.es1;
  g.p0 = s.base + (long long)(ix - DEG / 2) * s.es0 + (long long)(iy - DEG / 2) * s.es1;
  eu_accumulate<NCH, DEG>(s.wm, tx, ty, g, out);
}

// runtime-degree evaluator for degrees above the specialised ones
template <int NCH>
__device__ void eu_bspline_generic(const eu_src_dev &s, float cx, float cy, float *out)
{
  const int d = s.degree, order = d + 1;
  float gx = eu_gate(cx, s.gate0, s.lower0, s.upper0);
  float gy = eu_gate(cy, s.gate1, s.lower1, s.upper1);
  float fx = (d & 1) ? floorf(gx) : roundf(gx);
  float fy = (d & 1) ? floorf(gy) : roundf(gy);
  float tx = gx - fx, ty = gy - fy;
  const float *p = s.base + (long long)(int)fx * s.es0 + (long long)(int)fy * s.es1;
  float wx[EU_MAX_DEGREE + 1], wy[EU_MAX_DEGREE + 1];
  for (int c = 0; c <= d; c++) { wx[c] = s.wm[c * order]; wy[c] = wx[c]; }
  float px = tx, py = ty;
  for (int row = 1; row <= d; row++) {
    for (int c = 0; c <= d; c++) {
      wx[c] = wx[c] + px * s.wm[c * order + row];
      wy[c] = wy[c] + py * s.wm[c * order + row];
    }
    if (row < d) { px = px * tx; py = py * ty; }
  }
  const float *p0 = p - (d / 2) * s.es1 - (d / 2) * s.es0;
  float sum[NCH];
  for (int j = 0; j < order; j++) {
    const float *rowp = p0 + j * s.es1;
    for (int c = 0; c < NCH; c++) {
      float r = rowp[c] * wx[0];
      for (int i = 1; i < order; i++) r = r + wx[i] * rowp[i * s.es0 + c];
      if (j == 0) sum[c] = r * wy[0];
      else sum[c] = sum[c] + r * wy[j];
    }
  }
  for (int c = 0; c < NCH; c++) out[c] = sum[c];
}

// ---------------------------------------------------------------------------
// source lookup: ray -> source pixel coordinate
// ---------------------------------------------------------------------------

// ray_to_cubeface, geometry.h:1178-1289. The three dominance classes are
// mutually exclusive and exhaustive; when the whole wavefront agrees (ballot)
// only that class is evaluated - the GPU form of the reference's any_of()
// early-outs.
__device__ __forceinline__ void eu_cubeface(float rx, float ry, float rz,
                                            int &face, float &in0, float &in1)
{
  float ax = fabsf(rx), ay = fabsf(ry), az = fabsf(rz);
  bool m1 = ax >= ay, m2 = ax >= az, m3 = ay >= az;
  bool domx = m1 && m2, domz = (!m2) && (!m3);
  unsigned long long bx = __ballot(domx), bz = __ballot(domz);
  unsigned long long act = __ballot(1);
  if (bx == act) {
    face = rx < 0.0f ? 0 : 1;
    in0 = -rz / rx;
    in1 = ry / ax;
  } else if (bz == act) {
    face = rz < 0.0f ? 5 : 4;
    in0 = rx / rz;
    in1 = ry / az;
  } else if ((bx | bz) == 0ull) {
    face = ry < 0.0f ? 2 : 3;
    in0 = -rx / ay;
    in1 = rz / ry;
  } else {
    // mixed wavefront near a cube edge: select per lane
    float num0 = domx ? -rz : (domz ? rx : -rx);
    float den0 = domx ? rx : (domz ? rz : ay);
    float num1 = domx ? ry : (domz ? ry : rz);
    float den1 = domx ? ax : (domz ? az : ry);
    in0 = num0 / den0;
    in1 = num1 / den1;
    face = domx ? (rx < 0.0f ? 0 : 1) : (domz ? (rz < 0.0f ? 5 : 4) : (ry < 0.0f ? 2 : 3));
  }
}

// returns false for a miss (mount_t::get_coordinate mask, environment.h:1117-1149)
__device__ __forceinline__ bool eu_source_coordinate(const eu_src_dev &s, float rx,
                                                     float ry, float rz, float &sx,
                                                     float &sy, int &face)
{
  face = 0;
  if (s.prj == EU_CUBEMAP || s.prj == EU_BIATAN6) {
    float in0, in1;
    eu_cubeface(rx, ry, rz, face, in0, in1);
    if (s.prj == EU_BIATAN6) {
      const float k = (float)(4.0 / 3.14159265358979323846);
      in0 = k * eu_atanf(in0);
      in1 = k * eu_atanf(in1);
    }
    // cubemap_view_t::get_pickup_coordinate_px, environment.h:1452-1460
    float p0 = in0 + s.refc_md, p1 = in1 + s.refc_md;
    p0 = p0 * s.model_to_px;
    p1 = p1 * s.model_to_px;
    p1 = p1 + (float)(face * s.section_px);
    sx = p0 - .5f;
    sy = p1 - .5f;
    return true;
  }
  float c0, c1;
  switch (s.prj) {
    case EU_SPHERICAL: {       // ray_to_ll_t, geometry.h:278-301
      const float q2 = rx * rx + rz * rz;
      const float q = eu_sqrt2_guarded((eu_f2){ q2, q2 }).x;
      // the two atan2f as one packed evaluation (same bits as eu_atan2f, eu_math2.h)
      const eu_f2 a = eu_atan2f_2((eu_f2){ ry, rx }, (eu_f2){ q, rz });
      c1 = a.x;
      c0 = a.y;
      break;
    }
    case EU_CYLINDRICAL: {     // ray_to_cyl_t, geometry.h:389-410
      const float q2 = rx * rx + rz * rz;
      const float q = eu_sqrt2_guarded((eu_f2){ q2, q2 }).x;
      c1 = eu_div2_guarded((eu_f2){ ry, ry }, (eu_f2){ q, q }).x;
      c0 = eu_atan2f(rx, rz);
      break;
    }
    case EU_RECTILINEAR:       // ray_to_rect_t, geometry.h:328-345
    {
      const eu_f2 c = eu_div2_guarded((eu_f2){ rx, ry }, (eu_f2){ rz, rz });
      c0 = c.x;
      c1 = c.y;
      break;
    }
    case EU_STEREOGRAPHIC: {   // ray_to_ster_t, geometry.h:445-465
      float rn = 1.0f / sqrtf(rx * rx + ry * ry + rz * rz);
      float r = rx * rn, d = ry * rn, f = rz * rn;
      float factor = 2.0f / (f + 1.0f);
      c0 = r * factor;
      c1 = d * factor;
      break;
    }
    default: {                 // ray_to_fish_t, geometry.h:513-531
      const float q2 = rx * rx + ry * ry;
#if EU_COORD_LEAN
      const float q = eu_sqrt2_guarded((eu_f2){ q2, q2 }).x;     // sqrtf's bits, a quarter of its cycles
#else
      const float q = sqrtf(q2);
#endif
      const eu_f2 a = eu_atan2f_2((eu_f2){ rz, ry }, (eu_f2){ q, rx });
      float r = (float)1.57079632679489661923 - a.x;
      float phi = a.y;
      float sn, cs;
#if EU_COORD_SINCOS
      eu_sincosf_120(phi, &sn, &cs);     // |phi| <= pi: eu_cosf(phi), eu_sinf(phi) from one reduction
#else
      cs = eu_cosf(phi); sn = eu_sinf(phi);
#endif
      c0 = r * cs;
      c1 = r * sn;
      break;
    }
  }
  if (s.has_lcp) {
    // pto_planar, forward direction (environment.h:254-284): radial polynomial
    // (lens_correction.h:93-105), shift, shear
    float o0 = c0, o1 = c1;
    {
      float sqn = c0 * c0;
      sqn = sqn + c1 * c1;
#if EU_COORD_LEAN
      const float sq = eu_sqrt2_guarded((eu_f2){ sqn, sqn }).x;
      float x = eu_div2_guarded((eu_f2){ sq, sq }, (eu_f2){ s.lens_s, s.lens_s }).x;     // sqrtf(sqn) / s.lens_s
#else
      float x = sqrtf(sqn) / s.lens_s;
#endif
      float sum = 0.0f, power = 1.0f;
      sum = sum + s.lens_d * power; power = power * x;
      sum = sum + s.lens_c * power; power = power * x;
      sum = sum + s.lens_b * power; power = power * x;
      sum = sum + s.lens_a * power;
      o0 = o0 * sum; o1 = o1 * sum;
    }
    if (s.has_shift) { o0 = o0 + s.lens_h; o1 = o1 + s.lens_v; }
    if (s.has_shear) {
      float h0 = (float)((double)o0 + ((double)o1 * s.shear_g));
      float h1 = (float)((double)o1 + ((double)o0 * s.shear_t));
      o0 = h0; o1 = h1;
    }
    c0 = o0; c1 = o1;
  }
  // source_t::test_crd, environment.h:970-977 (float compares)
  bool mask = c0 >= s.wex0 && c0 <= s.wex1 && c1 >= s.wex2 && c1 <= s.wex3;
  if (s.prj == EU_RECTILINEAR) mask = mask && (rz > 0.0f);
  // source_t::md_to_spline, environment.h:988-1006: the subtraction is done in
  // double (vec<float> - double), everything after it in float
  // (x and y as one pair: packed operations, and the two divisions as one range-checked FMA sequence -
  // the bits of `/`, a quarter of its cycles)
  eu_f2 i = { (float)((double)c0 - s.tex_x0), (float)((double)c1 - s.tex_y0) };
#if EU_COORD_LEAN
  i = eu_div2_guarded(i, (eu_f2){ s.ext_w, s.ext_h });
#else
  i = i / (eu_f2){ s.ext_w, s.ext_h };
#endif
  i = i * (eu_f2){ s.total_w, s.total_h };
  i = i - .5f;
  i = i - (eu_f2){ s.win_x_off, s.win_y_off };
  sx = i.x;
  sy = i.y;
  return mask;
}

// repix_t, environment.h:1205-1309: channel-count adaption IN -> out_n
template <int IN>
__device__ __forceinline__ void eu_repix(int out_n, const float *in, float *out)
{
  if constexpr (IN == 1) {
    out[0] = in[0]; out[1] = in[0]; out[2] = in[0];
    if (out_n == 2) out[1] = 1.0f;
    if (out_n == 4) out[3] = 1.0f;
  } else if constexpr (IN == 2) {
    if (out_n == 4) { out[0] = in[0]; out[1] = in[0]; out[2] = in[0]; out[3] = in[1]; }
    else {
      float v = in[0] / in[1];
      if (in[1] == 0.0f) v = 0.0f;
      out[0] = v; out[1] = v; out[2] = v;
    }
  } else if constexpr (IN == 3) {
    if (out_n == 4) { out[0] = in[0]; out[1] = in[1]; out[2] = in[2]; out[3] = 1.0f; }
    else {
      float sum = in[0]; sum = sum + in[1]; sum = sum + in[2];
      out[0] = sum / 3.0f; out[1] = 1.0f;
    }
  } else {
    if (out_n == 1) {
      float v = (in[0] + in[1] + in[2]) / 3.0f;
      v = v / in[3];
      if (in[3] == 0.0f) v = 0.0f;
      out[0] = v;
    } else if (out_n == 2) {
      out[0] = (in[0] + in[1] + in[2]) / 3.0f; out[1] = in[3];
    } else {
      out[0] = in[0] / in[3]; out[1] = in[1] / in[3]; out[2] = in[2] / in[3];
      if (in[3] == 0.0f) { out[0] = 0.0f; out[1] = 0.0f; out[2] = 0.0f; }
    }
  }
}

template <int NCH, int DEG>
__device__ __forceinline__ void eu_environment(const eu_src_dev &s, float rx, float ry,
                                               float rz, float *px);

// --mask_for: the facet's inner evaluator is masking_t (1 or 3 channels: the paint,
// unconditionally) or alpha_masking_t (2 or 4: colour = paint * alpha, alpha kept), masking.h:70-135
template <int NCH>
__device__ __forceinline__ void eu_paint(int mask_paint, float *px)
{
  const float paint = mask_paint == 2 ? 1.0f : 0.0f;
  if constexpr (NCH == 1 || NCH == 3) {
#pragma unroll
    for (int c = 0; c < NCH; c++) px[c] = paint;
  } else {
    px[0] = paint * px[NCH - 1];
    if constexpr (NCH == 4) { px[1] = px[0]; px[2] = px[0]; }
  }
}

// mono_t, environment.h:1325-1383: the channel adaption of masking jobs, IN -> out_n in {1, 2}
// (the host refuses other combinations, as the reference asserts on them). out has room for 4.
template <int IN>
__device__ __forceinline__ void eu_mono(int out_n, const float *in, float *out)
{
  if constexpr (IN == 1) { out[0] = in[0]; out[1] = 1.0f; }
  else if constexpr (IN == 2) {
    float v = in[0] / in[1];
    if (in[1] == 0.0f) v = 0.0f;
    out[0] = v;
  } else if constexpr (IN == 3) { out[0] = in[0]; out[1] = 1.0f; }
  else {
    if (out_n == 1) {
      float v = in[0] / in[3];
      if (in[3] == 0.0f) v = 0.0f;
      out[0] = v;
    } else { out[0] = in[0]; out[1] = in[3]; }
  }
}

// environment::eval behind the coordinate stage, for callers that already have
// the source coordinate (the multi-facet synopsis computes it for the mask):
// inner evaluation, repix, brighten on the OUTPUT layout (environment.h:1821-1842,
// :1859-1900). px has room for 4 floats.
template <int NCH, int DEG>
__device__ __forceinline__ void eu_environment_repix_at(const eu_src_dev &s, int out_n, bool hit,
                                                        float sx, float sy, float *px)
{
  float raw[NCH];
  if (hit) {
    if constexpr (DEG >= 0) eu_bspline<NCH, DEG>(s, sx, sy, raw);
    else eu_bspline_generic<NCH>(s, sx, sy, raw);
    if (s.mask_paint) eu_paint<NCH>(s.mask_paint, raw);
  } else {
#pragma unroll
    for (int c = 0; c < NCH; c++) raw[c] = 0.0f;
  }
  if (s.mask_paint) eu_mono<NCH>(out_n, raw, px);      // environment.h:1909-1957
  else eu_repix<NCH>(out_n, raw, px);
  if (s.brighten != 1.0f) {
    const int ncol = (out_n == 2 || out_n == 4) ? out_n - 1 : out_n;
    for (int c = 0; c < ncol; c++) px[c] = px[c] * s.brighten;
  }
}

template <int NCH, int DEG>
__device__ __forceinline__ void eu_environment_repix(const eu_src_dev &s, int out_n, float rx,
                                                     float ry, float rz, float *px)
{
  float sx, sy;
  int face;
  const bool hit = eu_source_coordinate(s, rx, ry, rz, sx, sy, face);
  eu_environment_repix_at<NCH, DEG>(s, out_n, hit, sx, sy, px);
}

template <int NCH, int DEG>
__device__ __forceinline__ void eu_environment_at(const eu_src_dev &s, bool hit, float sx, float sy,
                                                  float *px)
{
  if (hit) {
    if constexpr (DEG >= 0) eu_bspline<NCH, DEG>(s, sx, sy, px);
    else eu_bspline_generic<NCH>(s, sx, sy, px);
    if (s.mask_paint) eu_paint<NCH>(s.mask_paint, px);
    // environment::eval, environment.h:1821-1842
    if (s.brighten != 1.0f) {
      constexpr int ncol = (NCH == 2 || NCH == 4) ? NCH - 1 : NCH;
#pragma unroll
      for (int c = 0; c < ncol; c++) px[c] = px[c] * s.brighten;
    }
  } else {
#pragma unroll
    for (int c = 0; c < NCH; c++) px[c] = 0.0f;
  }
}

template <int NCH, int DEG>
__device__ __forceinline__ void eu_environment(const eu_src_dev &s, float rx, float ry,
                                               float rz, float *px)
{
  float sx, sy;
  int face;
  const bool hit = eu_source_coordinate(s, rx, ry, rz, sx, sy, face);
  eu_environment_at<NCH, DEG>(s, hit, sx, sy, px);
}

// ---------------------------------------------------------------------------
// put side: zimt::storer + fluff (put.h:122-136), one pixel into its output row.
// The tethered pipeline's to_screen_t runs as a separate pass over the float
// frame (eu_setup.hip: to_screen_kernel) so that no render kernel carries it.
// ---------------------------------------------------------------------------

template <int NCH>
__device__ __forceinline__ void eu_put(float *row, int x, const float *px)
{
  float *o = row + (long long)x * NCH;
#pragma unroll
  for (int c = 0; c < NCH; c++) o[c] = px[c];
}

// ---------------------------------------------------------------------------
// target side: ray of pixel (x, y) from the stepper tables
// ---------------------------------------------------------------------------

__device__ __forceinline__ void eu_ray(int form, const float *rowt, float c0, float c1,
                                       float &rx, float &ry, float &rz)
{
  // rowt: A[3], B[3], C[3]
  if (form == EU_FORM_FISH || form == EU_FORM_STER) {
    // fisheye_stepper::work, stepper.h:1019-1030: a = M_PI_2 - norm(planar) is
    // formed in double and narrows to float before sin/cos (the float sincos
    // overload is the one that binds)
    const float p0 = c0, p1 = rowt[9];
    float sqn = p0 * p0;
    sqn = sqn + p1 * p1;
    const float nrm = sqrtf(sqn);
    // stereographic_stepper::work, stepper.h:1146: a = M_PI_2 - 2.0 * atan(norm / 2.0),
    // double atan, see eu_ster_angle
    const float a = form == EU_FORM_STER ? eu_ster_angle(nrm)
                                         : (float)(1.57079632679489661923 - (double)nrm);
    const float bb = eu_atan2f(p0, p1);
    const float z = eu_sinf(a), r = eu_cosf(a), sx = eu_sinf(bb), sy = eu_cosf(bb);
    rx = rowt[0] * r * sx + rowt[6] * z + rowt[3] * r * sy;
    ry = rowt[1] * r * sx + rowt[7] * z + rowt[4] * r * sy;
    rz = rowt[2] * r * sx + rowt[8] * z + rowt[5] * r * sy;
  } else if (form == EU_FORM_BCA) {
    rx = rowt[3] * c0 + rowt[6] * c1 + rowt[0];
    ry = rowt[4] * c0 + rowt[7] * c1 + rowt[1];
    rz = rowt[5] * c0 + rowt[8] * c1 + rowt[2];
  } else {
    rx = rowt[3] * c0 + rowt[0];
    ry = rowt[4] * c0 + rowt[1];
    rz = rowt[5] * c0 + rowt[2];
  }
}

__device__ __forceinline__ float eu_norm3(float x, float y, float z)
{
  // xel.h:752-765
  float sqn = x * x;
  sqn = sqn + y * y;
  sqn = sqn + z * z;
  return sqrtf(sqn);
}

// rotate(xel_t<float,3>, r3_t<float>), geometry.h:80-87
__device__ __forceinline__ void eu_rotate3(const float *m, float x, float y, float z, float &ox,
                                           float &oy, float &oz)
{
  const float a = x * m[0] + y * m[3], b = x * m[1] + y * m[4], c = x * m[2] + y * m[5];
  ox = a + z * m[6]; oy = b + z * m[7]; oz = c + z * m[8];
}

// generic_stepper's tf (tf_ex_facet::eval, envutil_payload.cc:1869-1884): planar -> ray by the target's
// projection (geometry.h: ll_to_ray_t :152-211, cyl_to_ray_t :417-446, rect_to_ray_t :363-387,
// ster_to_ray_t :481-510, fish_to_ray_t :539-566, ir_to_ray_t :663-770, ba6_to_ray_t :855-1000), then tf3d_t::eval (:1896-1941; its all_of / any_of
// tests only skip work)
__device__ __forceinline__ void eu_tf3d_eval(const eu_tf3d &g, float x, float y, float z, float &rx,
                                             float &ry, float &rz)
{
  if (!g.has_shift) { eu_rotate3(g.trg_to_src, x, y, z, rx, ry, rz); return; }
  float tx, ty, tz;
  eu_rotate3(g.trg_to_md, x, y, z, tx, ty, tz);
  const bool mask = tz <= 0.0f;
  tx = tx / tz; ty = ty / tz; tz = 1.0f;
  tx = tx * g.dcp; ty = ty * g.dcp; tz = tz * g.dcp;
  tx = tx - g.shift[0]; ty = ty - g.shift[1]; tz = tz - g.shift[2];
  eu_rotate3(g.md_to_src, tx, ty, tz, rx, ry, rz);
  if (mask) { rx = 0.0f; ry = 0.0f; rz = -__builtin_huge_valf(); }
}

// inverse_lcp::eval, lens_correction.h:289-299: the argument is a double (norm / s), the spline
// coordinate narrows to float at the evaluator; clamp gate of a NATURAL spline, cubic weights
__device__ __forceinline__ float eu_inv_lcp(const eu_inv_planar &q, double x)
{
  double t = x / q.rr_max;
  t = sqrt(t);
  t = t * (double)(q.nk - 1);
  const float c = (float)t, upper = (float)(q.nk - 1);
  float g = c;
  if (c < 0.0f) g = 0.0f;
  if (c > upper) g = upper;
  const float fl = floorf(g), delta = g - fl;
  float w[4];
  eu_weights<3>(q.m, delta, w);
  const float *p = q.coef + (int)fl - 1;
  float sum = p[0] * w[0];
  sum = sum + w[1] * p[1];
  sum = sum + w[2] * p[2];
  sum = sum + w[3] * p[3];
  return sum + 1.0f;
}

// Only kernels instantiated with GEN = true contain this path (eu_stepper<GEN>): inlined into the ordinary
// kernels - rare as it is, and with double arithmetic in it - it cost the multi-facet kernel 48 bytes of
// scratch per thread and 14 % of config 5; as a call it raised every kernel's register count to its own.
__device__ __forceinline__ void eu_generic_ray(const eu_generic &g, const eu_inv_planar *inv, float p0,
                                               float p1, float &rx, float &ry, float &rz)
{
  if (inv) {                  // pto_planar<T, L, true>::eval, environment.h:285-307
    if (inv->shear) {
      p1 = (float)(((double)p1 - inv->shear_t * (double)p0) / (1 - inv->shear_t * inv->shear_g));
      p0 = (float)((double)p0 - inv->shear_g * (double)p1);
    }
    if (inv->shift) { p0 = p0 - inv->h; p1 = p1 - inv->v; }
    if (inv->lcp) {
      float sqn = p0 * p0;
      sqn = sqn + p1 * p1;
      const float factor = eu_inv_lcp(*inv, (double)sqrtf(sqn) / inv->s);
      p0 = p0 * factor; p1 = p1 * factor;
    }
  }
  float x, y, z;
  if (g.prj == EU_SPHERICAL) {
    const float sinlat = eu_sinf(p1), coslat = eu_cosf(p1), sinlon = eu_sinf(p0), coslon = eu_cosf(p0);
    x = sinlon * coslat; z = coslon * coslat; y = sinlat;
  } else if (g.prj == EU_CYLINDRICAL) {
    z = eu_cosf(p0); x = eu_sinf(p0); y = p1;
  } else if (g.prj == EU_RECTILINEAR) {
    x = p0; y = p1; z = 1.0f;
  } else if (g.prj == EU_CUBEMAP || g.prj == EU_BIATAN6) {
    // ir_to_ray_t (geometry.h:663-770) / ba6_to_ray_t (:855-1000) as roll_out_23 default-constructs them:
    // section_md 2, refc_md 1
    float c0 = p0 + 1.0f, c1 = p1 + 6.0f;
    const int section = (int)((double)c1 / 2.0);
    c1 = (float)((double)c1 - (double)section * 2.0);
    c0 = c0 - 1.0f; c1 = c1 - 1.0f;
    if (g.prj == EU_BIATAN6) {
      const float q = (float)(3.14159265358979323846 / 4);
      c0 = eu_tanf(c0 * q); c1 = eu_tanf(c1 * q);
    }
    x = 0.0f; y = 0.0f; z = 0.0f;
    if (section == 1)      { x = 1.0f;  y = c1;    z = -c0; }
    else if (section == 0) { x = -1.0f; y = c1;    z = c0; }
    else if (section == 3) { x = -c0;   y = 1.0f;  z = c1; }
    else if (section == 2) { x = -c0;   y = -1.0f; z = -c1; }
    else if (section == 4) { x = c0;    y = c1;    z = 1.0f; }
    else if (section == 5) { x = -c0;   y = c1;    z = -1.0f; }
  } else {
    const float r = sqrtf(p0 * p0 + p1 * p1);
    const float theta = g.prj == EU_STEREOGRAPHIC ? eu_atanf(r / 2.0f) * 2.0f : r;
    const float phi = eu_atan2f(p0, -p1);
    const float st = eu_sinf(theta);
    z = eu_cosf(theta); y = -st * eu_cosf(phi); x = st * eu_sinf(phi);
  }
  eu_tf3d_eval(g.tf[0], x, y, z, rx, ry, rz);
  if (g.ntf == 2) { const float a = rx, b = ry, c = rz; eu_tf3d_eval(g.tf[1], a, b, c, rx, ry, rz); }
}

// full stepper: tables -> ray, with the normalisation flavour of the stepper. gen / raw: the facet's
// eu_generic and the raw planar x column, for form == EU_FORM_GENERIC
template <bool GEN = false>
__device__ __forceinline__ void eu_stepper(int form, int norm_mode, const float *colA,
                                           const float *colB, const float *rowt, int x,
                                           float &rx, float &ry, float &rz,
                                           const eu_generic *gen = nullptr, const float *raw = nullptr,
                                           const eu_inv_planar *inv = nullptr)
{
  if (GEN && form == EU_FORM_GENERIC) {
    if constexpr (GEN) eu_generic_ray(*gen, inv, raw[x], rowt[9], rx, ry, rz);
  }
  else eu_ray(form, rowt, colA[x], colB[x], rx, ry, rz);
  if (norm_mode == EU_NORM_DIV) {
    float n = eu_norm3(rx, ry, rz);
    rx = rx / n; ry = ry / n; rz = rz / n;
  } else if (norm_mode == EU_NORM_CYL) {
    // cylindrical_stepper keeps the reciprocal length of the lane's FIRST
    // pixel in the 512-pixel segment (stepper.h:771-775, :786)
    int seg = (x / EU_SEGMENT) * EU_SEGMENT;
    int x0 = seg + ((x - seg) % EU_LANES);
    float fx, fy, fz;
    eu_ray(form, rowt, colA[x0], colB[x0], fx, fy, fz);
    float rcp = 1.0f / eu_norm3(fx, fy, fz);
    rx = rx * rcp; ry = ry * rcp; rz = rz * rcp;
  }
}


// XCD-aware tile order. Workgroups b and b+8 share an XCD (round-robin
// dispatch). Tiles are grouped into units of unit_rows tile rows; unit u
// belongs to XCD u % 8 and an XCD walks its units in order: consecutive
// workgroups of one XCD are neighbouring tiles (they share source rows in
// that XCD's L2), while every XCD gets a slice of every part of the frame
// (cube faces differ in cost). Returns the tile index or -1 for the padding
// workgroups of the last round; launch eu_xcd_grid() workgroups.
__device__ __forceinline__ int eu_xcd_tile(int blk, int tiles_x, int tiles_y, int unit_rows)
{
  const int nx = 8;
  const int xcd = blk % nx, k = blk / nx;
  if (unit_rows < 0) {
    // unit_rows < 0: inside a unit of -unit_rows tile rows the tiles are walked COLUMN by column (down the
    // unit's tile rows, then one tile to the right). What an XCD has in flight is then a compact block of the
    // frame, and the source lines two vertically neighbouring tiles share - under a rotated or otherwise
    // slanted mapping most of them - are still in its L2 when the second tile asks: config 4 (32K -> 32K
    // rotated, twined) FETCH_SIZE x 2 = 12.06 -> 8.26 GB for a 6.44 GB source, HBM traffic 1.44 -> 1.14 times
    // the algorithmic bytes; config 5 8.25 -> 5.76 GB read and 7.64 -> 7.43 ms. Upright lat/lon jobs keep the
    // row-major walk (config 2: 0.228 ms row-major, 0.232 column-major).
    const int ur = -unit_rows;
    const int unit_tiles = ur * tiles_x;
    const int ul = k / unit_tiles, iu = k - ul * unit_tiles;
    const int ix = iu / ur, iy = iu - ix * ur;
    const int ty = (ul * nx + xcd) * ur + iy;
    return ty < tiles_y ? ty * tiles_x + ix : -1;
  }
  const int unit_tiles = unit_rows * tiles_x;
  const int ul = k / unit_tiles, iu = k - ul * unit_tiles;
  const int b = (ul * nx + xcd) * unit_tiles + iu;
  return b < tiles_x * tiles_y ? b : -1;
}

static inline int eu_xcd_grid(int tiles_x, int tiles_y, int unit_rows)
{
  if (unit_rows < 0) unit_rows = -unit_rows;
  const int units = (tiles_y + unit_rows - 1) / unit_rows;
  return ((units + 7) / 8) * 8 * unit_rows * tiles_x;
}

#define EU_UNIT_ROWS 8

template <bool GEN = false>
__device__ __forceinline__ void eu_stepper(const eu_render_params &p, const float *colA,
                                           const float *colB, const float *rowt, int x,
                                           float &rx, float &ry, float &rz)
{
  if constexpr (GEN) {
    // colA is p.col (r00, r01) or p.col + 2 * width (r10): the raw planar column goes with it
    const float *raw = p.col + (colA == p.col ? 4 : 5) * (long long)p.width;
    eu_stepper<true>(p.form, p.norm_mode, colA, colB, rowt, x, rx, ry, rz, &p.gen, raw,
                     (p.inv.shear | p.inv.shift | p.inv.lcp) ? &p.inv : nullptr);
  } else {
    eu_stepper<false>(p.form, p.norm_mode, colA, colB, rowt, x, rx, ry, rz);
  }
}

#endif

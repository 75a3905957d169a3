// Packed render kernel: TWO output pixels per lane (x and x+64 of one row),
// all coordinate arithmetic on float2 so that it issues as v_pk_* instructions
// (see eu_math2.h), divisions/square roots by the range-checked FMA sequences,
// b-spline weights with the structural zeros of the weight matrix skipped.
// Same operations, same order, same bits as eu_render_kernel - only how they
// are issued changes. Covers the jobs without twining whose source is a
// lat/lon image; everything else stays on eu_render_kernel.
#include <cstdlib>
#include "eu_render_dev.h"
#include "eu_math2.h"

#define EU2_TILE_W 128   // pixels of one row per wave and pass (2 per lane)
#define EU2_TILE_H 4   // default waves (rows) per workgroup
#define EU2_UNIT_ROWS 8  // tile rows per XCD unit

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
    eu_f2 w0 = a + d * -0.5f; w0 = w0 + d2 * 0.5f; w0 = w0 + d3 * -a;
    eu_f2 w1 = b - d2;        w1 = w1 + d3 * 0.5f;
    eu_f2 w2 = a + d * 0.5f;  w2 = w2 + d2 * 0.5f; w2 = w2 + d3 * -0.5f;
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
      sum = sum + b[c] * wr0;
      sum = sum * wl1;
      float sub = c2[c] * wl0;
      sub = sub + d[c] * wr0;
      sum = sum + sub * wr1;
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
        for (int i = 1; i < order; i++) r = r + wx[i] * t[i][c];
        if (j == 0) sum[c] = r * wy[0];
        else sum[c] = sum[c] + r * wy[j];
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
      i0 = i0 / s.ext_w;
      i1 = i1 / s.ext_h;
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
// the kernel
// ---------------------------------------------------------------------------

// EU2_WAVES (build-time experiment): cap the registers for that many waves per SIMD
#ifdef EU2_WAVES
#define EU2_OCC __attribute__((amdgpu_waves_per_eu(EU2_WAVES, EU2_WAVES)))
#else
#define EU2_OCC
#endif

template <int NCH, int DEG, int PRJ, bool TWINE>
__global__ __launch_bounds__(256) EU2_OCC void eu_render2_kernel(const eu_render_params p)
{
  // atanf range table in LDS (eu_math2.h): filled before any thread leaves
  __shared__ __attribute__((aligned(16))) float atab[EU_ATAN_TAB_FLOATS];
  if constexpr (PRJ != EU_CUBEMAP) {
    if (threadIdx.x < EU_ATAN_TAB_ENTRIES) eu_atan_tab_entry(threadIdx.x, atab + 8 * threadIdx.x);
    __syncthreads();
  }
  const int b = eu_xcd_tile(blockIdx.x, p.tiles_x, p.tiles_y, p.unit_rows);
  if (b < 0) return;
  const int tile_y = b / p.tiles_x, tile_x = b - tile_y * p.tiles_x;
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int y = p.row_begin + tile_y * EU2_TILE_H + wave;
  if (y >= p.row_end) return;
  const int xa = tile_x * EU2_TILE_W + lane, xb = xa + 64;
  if (xa >= p.width) return;
  const bool vb = xb < p.width;
  const int xbc = vb ? xb : xa;
  const eu_src_dev &s = p.src;

  eu_cptr rowt = (eu_cptr)(p.row + (long long)eu_frame_row(y, p.band_shift, p.band_count, p.band_index) * EU_ROW_FLOATS);
  const eu_ray2 r00 = eu_rays2(p.form, p.norm_mode, rowt, p.col, p.col + p.width, xa, xbc);

  float pxa[NCH], pxb[NCH];
  if constexpr (!TWINE) {
    eu_f2 sx, sy;
    const eu_i2 hit = eu_coord2<PRJ>(s, r00, sx, sy, atab);
    eu_eval2<NCH, DEG>(s, sx, sy, hit, pxa, pxb);
  } else {
    // deriv_stepper (stepper.h:1591-1715) + twine_t::eval (twining.h:128-263)
    const eu_ray2 r10 = eu_rays2(p.form, p.norm_mode, rowt, p.col + 2 * p.width,
                                 p.col + 3 * p.width, xa, xbc);
    const eu_ray2 r01 = eu_rays2(p.form, p.norm_mode, rowt + EU_ROW_VARIANT, p.col, p.col + p.width, xa, xbc);
    const eu_f2 dxx = r10.x - r00.x, dxy = r10.y - r00.y, dxz = r10.z - r00.z;
    const eu_f2 dyx = r01.x - r00.x, dyy = r01.y - r00.y, dyz = r01.z - r00.z;
#pragma unroll
    for (int c = 0; c < NCH; c++) { pxa[c] = 0.0f; pxb[c] = 0.0f; }
    eu_cptr taps = (eu_cptr)p.taps;
    for (int k = 0; k < p.ntaps; k++) {
      const float cx = taps[3 * k], cy = taps[3 * k + 1], cw = taps[3 * k + 2];
      eu_ray2 rk;
      rk.x = r00.x + cx * dxx + cy * dyx;
      rk.y = r00.y + cx * dxy + cy * dyy;
      rk.z = r00.z + cx * dxz + cy * dyz;
      eu_f2 sx, sy;
      const eu_i2 hit = eu_coord2<PRJ>(s, rk, sx, sy, atab);
      float qa[NCH], qb[NCH];
      eu_eval2<NCH, DEG>(s, sx, sy, hit, qa, qb);
#pragma unroll
      for (int c = 0; c < NCH; c++) { pxa[c] = pxa[c] + cw * qa[c]; pxb[c] = pxb[c] + cw * qb[c]; }
    }
  }

  float *o = p.out + (long long)(y - p.row_begin) * p.out_stride;
  eu_put<NCH>(o, xa, pxa);
  if (vb) eu_put<NCH>(o, xb, pxb);
}

// ---------------------------------------------------------------------------
// LDS-staged variant (no twining): 32x16 output tiles, still two pixels per
// lane (rows y and y+8 of one column), so that the tile's source footprint is a
// compact box for ANY orientation of the mapping (polar cube faces, rotated
// targets). The box is copied into LDS once with coalesced row reads - every
// source texel passes the texture addresser once per tile instead of once per
// tap - and the (d+1)^2 taps are 16-byte LDS reads. Tiles whose box exceeds the
// LDS budget (pole of a lat/lon source, the +-180 degree seam, strong
// minification) gather from global memory like eu_render2_kernel; the choice is
// per workgroup.
// ---------------------------------------------------------------------------

#define EU3_TW 32
#define EU3_TH 16
#define EU3_LDS_BYTES (36 * 1024)

__device__ __forceinline__ int eu_wmin(int v)
{
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = min(v, __shfl_xor(v, o));
  return v;
}
__device__ __forceinline__ int eu_wmax(int v)
{
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = max(v, __shfl_xor(v, o));
  return v;
}

template <int NCH, int DEG, int PRJ, bool STAGE>
__global__ __launch_bounds__(256, 4) void eu_render3_kernel(const eu_render_params p)
{
  constexpr int TEX = NCH == 3 ? 4 : NCH;              // floats per LDS texel
  constexpr int CAP = STAGE ? EU3_LDS_BYTES / (TEX * 4) : 4;   // texels
  __shared__ __attribute__((aligned(16))) float tile[CAP * TEX];
  __shared__ __attribute__((aligned(16))) float atab[EU_ATAN_TAB_FLOATS];
  __shared__ int bbw[4][4];
  if constexpr (PRJ != EU_CUBEMAP) {
    if (threadIdx.x < EU_ATAN_TAB_ENTRIES) eu_atan_tab_entry(threadIdx.x, atab + 8 * threadIdx.x);
  }
  const int b = eu_xcd_tile(blockIdx.x, p.tiles_x, p.tiles_y, p.unit_rows);
  if (b < 0) return;                                   // whole workgroup
  __syncthreads();
  const int tile_y = b / p.tiles_x, tile_x = b - tile_y * p.tiles_x;
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  // lane -> pixel inside the 32x16 tile. Staged: rows of 32. Direct gathers: the
  // four lanes the memory pipe serves together (a quad) form a 2x2 pixel block,
  // which touches about the same number of cache lines for every orientation of
  // the mapping; a wave covers 32x2 pixels either way.
  int lx, ly;
  if constexpr (STAGE) { lx = threadIdx.x & 31; ly = threadIdx.x >> 5; }
  else {
    const int l = threadIdx.x & 63;
    lx = ((l >> 2) << 1) | (l & 1);
    ly = ((threadIdx.x >> 6) << 1) | ((l >> 1) & 1);
  }
  const int x = tile_x * EU3_TW + lx;
  const int ya = p.row_begin + tile_y * EU3_TH + ly, yb = ya + 8;
  const bool la = x < p.width && ya < p.row_end, lb = x < p.width && yb < p.row_end;
  const int xc = x < p.width ? x : p.width - 1;
  const int yac = ya < p.row_end ? ya : p.row_end - 1, ybc = yb < p.row_end ? yb : p.row_end - 1;
  const eu_src_dev &s = p.src;

  // rays: the two pixels share the column value and differ in the row constants
  eu_ray2 r;
  {
    const float *ra = p.row + (long long)eu_frame_row(yac, p.band_shift, p.band_count, p.band_index) * EU_ROW_FLOATS;
    const float *rb = p.row + (long long)eu_frame_row(ybc, p.band_shift, p.band_count, p.band_index) * EU_ROW_FLOATS;
    const float c0 = p.col[xc];
    const eu_f2 A0 = { ra[0], rb[0] }, A1 = { ra[1], rb[1] }, A2 = { ra[2], rb[2] };
    const eu_f2 B0 = { ra[3], rb[3] }, B1 = { ra[4], rb[4] }, B2 = { ra[5], rb[5] };
    if (p.form == EU_FORM_BCA) {
      const float c1 = p.col[p.width + xc];
      const eu_f2 C0 = { ra[6], rb[6] }, C1 = { ra[7], rb[7] }, C2 = { ra[8], rb[8] };
      r.x = B0 * c0 + C0 * c1 + A0;
      r.y = B1 * c0 + C1 * c1 + A1;
      r.z = B2 * c0 + C2 * c1 + A2;
    } else {
      r.x = B0 * c0 + A0;
      r.y = B1 * c0 + A1;
      r.z = B2 * c0 + A2;
    }
  }
  eu_f2 sx, sy;
  eu_i2 hit = eu_coord2<PRJ>(s, r, sx, sy, atab);
  hit = hit & (eu_i2){ la ? -1 : 0, lb ? -1 : 0 };

  // gate + split
  eu_f2 gx = eu_gate2(sx, s.gate0, s.lower0, s.upper0);
  eu_f2 gy = eu_gate2(sy, s.gate1, s.lower1, s.upper1);
  eu_f2 fx, fy;
  if constexpr (DEG & 1) {
    fx = (eu_f2){ floorf(gx.x), floorf(gx.y) }; fy = (eu_f2){ floorf(gy.x), floorf(gy.y) };
  } else {
    fx = (eu_f2){ roundf(gx.x), roundf(gx.y) }; fy = (eu_f2){ roundf(gy.x), roundf(gy.y) };
  }
  const eu_f2 tx = gx - fx, ty = gy - fy;
  // lanes without a hit (misses, pixels outside the frame) use the window at
  // the core origin: inside every container, framed or not
  const int ixa = hit.x ? (int)fx.x : DEG / 2, iya = hit.x ? (int)fy.x : DEG / 2;
  const int ixb = hit.y ? (int)fx.y : DEG / 2, iyb = hit.y ? (int)fy.y : DEG / 2;

  // bounding box of the base positions of all hitting pixels of the tile
  bool fits = false;
  int mnx = INT_MAX, mny = INT_MAX, mxx = INT_MIN, mxy = INT_MIN;
  int bx0 = 0, by0 = 0;
  long long bw = 0, bh = 0;
  if constexpr (STAGE) {
  if (hit.x) { mnx = ixa; mxx = ixa; mny = iya; mxy = iya; }
  if (hit.y) { mnx = min(mnx, ixb); mxx = max(mxx, ixb); mny = min(mny, iyb); mxy = max(mxy, iyb); }
  mnx = eu_wmin(mnx); mny = eu_wmin(mny); mxx = eu_wmax(mxx); mxy = eu_wmax(mxy);
  if (lane == 0) { bbw[wave][0] = mnx; bbw[wave][1] = mny; bbw[wave][2] = mxx; bbw[wave][3] = mxy; }
  __syncthreads();
  mnx = min(min(bbw[0][0], bbw[1][0]), min(bbw[2][0], bbw[3][0]));
  mny = min(min(bbw[0][1], bbw[1][1]), min(bbw[2][1], bbw[3][1]));
  mxx = max(max(bbw[0][2], bbw[1][2]), max(bbw[2][2], bbw[3][2]));
  mxy = max(max(bbw[0][3], bbw[1][3]), max(bbw[2][3], bbw[3][3]));
  mnx = __builtin_amdgcn_readfirstlane(mnx); mny = __builtin_amdgcn_readfirstlane(mny);
  mxx = __builtin_amdgcn_readfirstlane(mxx); mxy = __builtin_amdgcn_readfirstlane(mxy);
  const bool any = mnx != INT_MAX;
  bx0 = mnx - DEG / 2; by0 = mny - DEG / 2;
  bw = (long long)mxx - mnx + DEG + 1; bh = (long long)mxy - mny + DEG + 1;
  fits = any && bw * bh <= CAP;
  }

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

  float pxa[NCH], pxb[NCH];
  if (fits) {
    const int ibw = (int)bw, ibh = (int)bh;
    // stage the box: wave w copies rows w, w+4, ...; lanes run along x
    for (int rr = wave; rr < ibh; rr += 4) {
      const float *g = s.base + (long long)(by0 + rr) * s.es1 + (long long)bx0 * NCH;
      for (int c = lane; c < ibw; c += 64) {
        const float *q = g + (long long)c * NCH;
        float *d = tile + (rr * ibw + c) * TEX;
        if constexpr (NCH == 3) {
          float v0 = q[0], v1 = q[1], v2 = q[2];
          *reinterpret_cast<float4 *>(d) = make_float4(v0, v1, v2, 0.0f);
        } else if constexpr (NCH == 4) {
          *reinterpret_cast<float4 *>(d) = *reinterpret_cast<const float4 *>(q);
        } else if constexpr (NCH == 2) {
          *reinterpret_cast<float2 *>(d) = *reinterpret_cast<const float2 *>(q);
        } else {
          d[0] = q[0];
        }
      }
    }
    __syncthreads();
    const int pitch = ibw * TEX;
    // lanes without a hit read the box origin (their result is discarded)
    const int oa = hit.x ? ((iya - mny) * ibw + (ixa - mnx)) * TEX : 0;
    const int ob = hit.y ? ((iyb - mny) * ibw + (ixb - mnx)) * TEX : 0;
    eu_lptr lt = (eu_lptr)tile;
    eu_accumulate1<NCH, DEG, TEX, int, eu_lptr>(lt + oa, pitch, wxa, wya, tx.x, ty.x, pxa);
    eu_accumulate1<NCH, DEG, TEX, int, eu_lptr>(lt + ob, pitch, wxb, wyb, tx.y, ty.y, pxb);
  } else {
    const float *pa = s.base + (long long)(ixa - DEG / 2) * NCH + (long long)(iya - DEG / 2) * s.es1;
    const float *pb = s.base + (long long)(ixb - DEG / 2) * NCH + (long long)(iyb - DEG / 2) * s.es1;
    eu_accumulate1<NCH, DEG>(pa, s.es1, wxa, wya, tx.x, ty.x, pxa);
    eu_accumulate1<NCH, DEG>(pb, s.es1, wxb, wyb, tx.y, ty.y, pxb);
  }
  constexpr int ncol = (NCH == 2 || NCH == 4) ? NCH - 1 : NCH;
  const bool bright = s.brighten != 1.0f;
  if (la) {
    float *o = p.out + (long long)(ya - p.row_begin) * p.out_stride + (long long)x * NCH;
#pragma unroll
    for (int c = 0; c < NCH; c++) {
      float v = pxa[c];
      if (bright && c < ncol) v = v * s.brighten;
      o[c] = hit.x ? v : 0.0f;
    }
  }
  if (lb) {
    float *o = p.out + (long long)(yb - p.row_begin) * p.out_stride + (long long)x * NCH;
#pragma unroll
    for (int c = 0; c < NCH; c++) {
      float v = pxb[c];
      if (bright && c < ncol) v = v * s.brighten;
      o[c] = hit.y ? v : 0.0f;
    }
  }
}


template <int NCH, int DEG, int PRJ>
static int launch2_ndp(const eu_render_params &p, hipStream_t st)
{
  // EU_HIP_LDS: 0 = row-strip tiles (eu_render2_kernel), 1 = 32x16 tiles staged
  // through LDS, 2 = 32x16 tiles with direct gathers
  static const int env_lds = [] { const char *e = getenv("EU_HIP_LDS"); return e ? atoi(e) : 0; }();
  const int use_lds = p.layout == 1 ? 0 : p.layout == 2 ? 2 : env_lds;
  if (!p.twine && use_lds && p.norm_mode == EU_NORM_NONE) {
    eu_render_params q = p;
    q.tiles_x = (p.width + EU3_TW - 1) / EU3_TW;
    q.tiles_y = (p.row_end - p.row_begin + EU3_TH - 1) / EU3_TH;
    static const int unit3 = [] { const char *e = getenv("EU_HIP_UNIT3"); return e ? atoi(e) : 2; }();
    q.unit_rows = unit3 > 0 ? unit3 : 2;
    dim3 grid3((unsigned)eu_xcd_grid(q.tiles_x, q.tiles_y, q.unit_rows)), block3(256);
    if (use_lds == 1) hipLaunchKernelGGL((eu_render3_kernel<NCH, DEG, PRJ, true>), grid3, block3, 0, st, q);
    else hipLaunchKernelGGL((eu_render3_kernel<NCH, DEG, PRJ, false>), grid3, block3, 0, st, q);
    return hipGetLastError() == hipSuccess ? 0 : -1;
  }
  dim3 grid((unsigned)eu_xcd_grid(p.tiles_x, p.tiles_y, p.unit_rows)), block(256);
  if (p.twine) hipLaunchKernelGGL((eu_render2_kernel<NCH, DEG, PRJ, true>), grid, block, 0, st, p);
  else hipLaunchKernelGGL((eu_render2_kernel<NCH, DEG, PRJ, false>), grid, block, 0, st, p);
  return hipGetLastError() == hipSuccess ? 0 : -1;
}

template <int NCH, int DEG>
static int launch2_nd(const eu_render_params &p, hipStream_t st)
{
  switch (p.src.prj) {
    case EU_SPHERICAL: return launch2_ndp<NCH, DEG, EU_SPHERICAL>(p, st);
    case EU_CUBEMAP: return launch2_ndp<NCH, DEG, EU_CUBEMAP>(p, st);
    case EU_BIATAN6: return launch2_ndp<NCH, DEG, EU_BIATAN6>(p, st);
  }
  return 1;
}

template <int NCH>
static int launch2_n(const eu_render_params &p, hipStream_t st)
{
  switch (p.src.degree) {
    case 1: return launch2_nd<NCH, 1>(p, st);
    case 2: return launch2_nd<NCH, 2>(p, st);
    case 3: return launch2_nd<NCH, 3>(p, st);
  }
  return 1;
}

// returns 1 when the job is outside this kernel's coverage (caller falls back
// to eu_render_kernel)
extern "C" int eu_launch_render2(const eu_render_params *pp, void *stream)
{
  eu_render_params p = *pp;
  if (p.stage != 0 || p.form >= EU_FORM_FISH || p.src.has_lcp || p.nch_out != p.nch) return 1;
  if (p.src.prj != EU_SPHERICAL && p.src.prj != EU_CUBEMAP && p.src.prj != EU_BIATAN6) return 1;
  if (p.src.degree < 1 || p.src.degree > 3 || p.src.es0 != p.nch) return 1;
  static const int unit_rows = [] { const char *e = getenv("EU_HIP_UNIT"); return e ? atoi(e) : EU2_UNIT_ROWS; }();
  p.unit_rows = unit_rows > 0 ? unit_rows : EU2_UNIT_ROWS;
  p.tiles_x = (p.width + EU2_TILE_W - 1) / EU2_TILE_W;
  p.tiles_y = (p.row_end - p.row_begin + EU2_TILE_H - 1) / EU2_TILE_H;
  if (p.tiles_x <= 0 || p.tiles_y <= 0) return 0;
  hipStream_t st = (hipStream_t)stream;
  switch (p.nch) {
    case 1: return launch2_n<1>(p, st);
    case 2: return launch2_n<2>(p, st);
    case 3: return launch2_n<3>(p, st);
    case 4: return launch2_n<4>(p, st);
  }
  return 1;
}

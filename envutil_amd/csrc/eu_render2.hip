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

// weighted sum for one pixel: channels 0/1 packed, the rest scalar
template <int NCH, int DEG>
__device__ __forceinline__ void eu_accumulate1(const float *__restrict__ p0, long long es1,
                                               const float *wx, const float *wy, float tx,
                                               float ty, float *out)
{
  if constexpr (DEG == 1) {
    float wl0 = 1.0f - tx, wr0 = tx, wl1 = 1.0f - ty, wr1 = ty;
    const float *q = p0 + es1;
    float a[NCH], b[NCH], c2[NCH], d[NCH];
#pragma unroll
    for (int c = 0; c < NCH; c++) { a[c] = p0[c]; b[c] = p0[NCH + c]; c2[c] = q[c]; d[c] = q[NCH + c]; }
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
      const float *rowp = p0 + j * es1;
      float t[order][NCH];
#pragma unroll
      for (int i = 0; i < order; i++)
#pragma unroll
        for (int c = 0; c < NCH; c++) t[i][c] = rowp[i * NCH + c];
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

template <int NCH, int DEG, int ROWS, bool PERSIST, int PASSES>
__global__ __launch_bounds__(64 * ROWS) void eu_render2_kernel(const eu_render_params p)
{
  // XCD-aware tile order. Workgroups b and b+8 share an XCD (round-robin
  // dispatch). Tiles are grouped into units of EU2_UNIT_ROWS tile rows; unit u
  // belongs to XCD u % 8, and an XCD walks its units in order. Inside a unit
  // consecutive workgroups of one XCD are neighbouring tiles (L2 reuse of the
  // source rows they share); across the frame every XCD gets a slice of every
  // cube face, so the slower polar faces do not all land on two XCDs.
  const int nblk = p.tiles_x * p.tiles_y;
  const int nx = 8;
  const int xcd = blockIdx.x % nx, kblk = blockIdx.x / nx;
  const int unit_tiles = p.unit_rows * p.tiles_x;
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  {
  const int ul = kblk / unit_tiles, iu = kblk - ul * unit_tiles;
  const int b = (ul * nx + xcd) * unit_tiles + iu;
  if (b >= nblk) return;
  const int tile_y = b / p.tiles_x, tile_x = b - tile_y * p.tiles_x;
  const int y = p.row_begin + tile_y * ROWS + wave;
  if (y >= p.row_end) return;
#pragma unroll 1
  for (int pass = 0; pass < PASSES; pass++) {
  const int xa = (tile_x * PASSES + pass) * EU2_TILE_W + lane, xb = xa + 64;
  if (xa >= p.width) continue;
  const bool vb = xb < p.width;
  const int xbc = vb ? xb : xa;

  // stepper tables -> rays (stepper.h; eu_setup_math.h build_stepper_tables)
  eu_cptr rowt = (eu_cptr)(p.row + (long long)y * EU_ROW_FLOATS);
  const float A0 = rowt[0], A1 = rowt[1], A2 = rowt[2], B0 = rowt[3], B1 = rowt[4], B2 = rowt[5];
  const eu_f2 c0 = { p.col[xa], p.col[xbc] };
  eu_f2 rx, ry, rz;
  if (p.form == EU_FORM_BCA) {
    const float C0 = rowt[6], C1 = rowt[7], C2 = rowt[8];
    const eu_f2 c1 = { p.col[p.width + xa], p.col[p.width + xbc] };
    rx = B0 * c0 + C0 * c1 + A0;
    ry = B1 * c0 + C1 * c1 + A1;
    rz = B2 * c0 + C2 * c1 + A2;
  } else {
    rx = B0 * c0 + A0;
    ry = B1 * c0 + A1;
    rz = B2 * c0 + A2;
  }

  // ray_to_ll_t (geometry.h:278-301): s = sqrt(r*r + f*f); lat = atan2(d, s); lon = atan2(r, f)
  const eu_src_dev &s = p.src;
  eu_f2 q2 = rx * rx + rz * rz;
  eu_f2 qs;
  {
    const eu_u2 iq = eu_bits2(q2);
    const eu_i2 ok = (iq - 0x2b800000u) <= (0x53800000u - 0x2b800000u);
    qs = eu_sqrt2_safe(eu_sel2(ok, q2, (eu_f2){ 1.0f, 1.0f }));
    if (__builtin_expect(!(ok.x & ok.y), 0)) {
      if (!ok.x) qs.x = sqrtf(q2.x);
      if (!ok.y) qs.y = sqrtf(q2.y);
    }
  }
  eu_f2 lat = eu_atan2f_2_xpos(ry, qs);
  eu_f2 lon = eu_atan2f_2(rx, rz);

  // mount_t::get_coordinate mask (environment.h:1117-1149)
  // a full-sphere image covers atan2f's whole range [-pi_f, pi_f] x [-pi_f/2, pi_f/2]
  // (the window extents narrow to exactly those floats): every ray hits
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
  eu_f2 sx = i0 - s.win_x_off, sy = i1 - s.win_y_off;

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

  const float *pa = s.base + (long long)((int)fx.x - DEG / 2) * NCH + (long long)((int)fy.x - DEG / 2) * s.es1;
  const float *pb = s.base + (long long)((int)fx.y - DEG / 2) * NCH + (long long)((int)fy.y - DEG / 2) * s.es1;
  float pxa[NCH], pxb[NCH];
  eu_accumulate1<NCH, DEG>(pa, s.es1, wxa, wya, tx.x, ty.x, pxa);
  eu_accumulate1<NCH, DEG>(pb, s.es1, wxb, wyb, tx.y, ty.y, pxb);

  constexpr int ncol = (NCH == 2 || NCH == 4) ? NCH - 1 : NCH;
  const bool bright = s.brighten != 1.0f;
  float *o = p.out + (long long)(y - p.row_begin) * p.out_stride;
#pragma unroll
  for (int c = 0; c < NCH; c++) {
    float v = pxa[c];
    if (bright && c < ncol) v = v * s.brighten;
    o[(long long)xa * NCH + c] = hit.x ? v : 0.0f;
  }
  if (vb) {
#pragma unroll
    for (int c = 0; c < NCH; c++) {
      float v = pxb[c];
      if (bright && c < ncol) v = v * s.brighten;
      o[(long long)xb * NCH + c] = hit.y ? v : 0.0f;
    }
  }
  }  // passes
  }
}

template <int NCH, int ROWS, bool PERSIST, int PASSES>
static int launch2_nr(const eu_render_params &p, hipStream_t st)
{
  const int unit_tiles = p.unit_rows * p.tiles_x;
  const int units = (p.tiles_y + p.unit_rows - 1) / p.unit_rows;
  int g = ((units + 7) / 8) * 8 * unit_tiles;
  dim3 grid((unsigned)g), block(64 * ROWS);
  switch (p.src.degree) {
    case 1: hipLaunchKernelGGL((eu_render2_kernel<NCH, 1, ROWS, PERSIST, PASSES>), grid, block, 0, st, p); break;
    case 2: hipLaunchKernelGGL((eu_render2_kernel<NCH, 2, ROWS, PERSIST, PASSES>), grid, block, 0, st, p); break;
    case 3: hipLaunchKernelGGL((eu_render2_kernel<NCH, 3, ROWS, PERSIST, PASSES>), grid, block, 0, st, p); break;
    default: return 1;
  }
  return hipGetLastError() == hipSuccess ? 0 : -1;
}

template <int NCH>
static int launch2_n(eu_render_params &p, hipStream_t st)
{
  static const int rows = [] { const char *e = getenv("EU_HIP_ROWS"); return e ? atoi(e) : EU2_TILE_H; }();
  static const int persist = [] { const char *e = getenv("EU_HIP_PERSIST"); return e ? atoi(e) : 0; }();
  static const int unit_rows = [] { const char *e = getenv("EU_HIP_UNIT"); return e ? atoi(e) : EU2_UNIT_ROWS; }();
  p.unit_rows = unit_rows > 0 ? unit_rows : EU2_UNIT_ROWS;
  static const int passes = [] { const char *e = getenv("EU_HIP_PASSES"); return e ? atoi(e) : 1; }();
  p.tiles_y = (p.row_end - p.row_begin + rows - 1) / rows;
  if (passes == 2) {
    p.tiles_x = (p.width + 2 * EU2_TILE_W - 1) / (2 * EU2_TILE_W);
    if (rows == 8) return launch2_nr<NCH, 8, false, 2>(p, st);
    p.tiles_y = (p.row_end - p.row_begin + 3) / 4;
    return launch2_nr<NCH, 4, false, 2>(p, st);
  }
  if (passes == 4) {
    p.tiles_x = (p.width + 4 * EU2_TILE_W - 1) / (4 * EU2_TILE_W);
    p.tiles_y = (p.row_end - p.row_begin + 3) / 4;
    return launch2_nr<NCH, 4, false, 4>(p, st);
  }
  switch (rows) {
    case 8: return launch2_nr<NCH, 8, false, 1>(p, st);
    case 16: return launch2_nr<NCH, 16, false, 1>(p, st);
    default: p.tiles_y = (p.row_end - p.row_begin + 3) / 4; return launch2_nr<NCH, 4, false, 1>(p, st);
  }
}

// returns 1 when the job is outside this kernel's coverage (caller falls back)
extern "C" int eu_launch_render2(const eu_render_params *pp, void *stream)
{
  eu_render_params p = *pp;
  if (p.twine || p.stage != 0 || p.norm_mode != EU_NORM_NONE || p.src.prj != EU_SPHERICAL) return 1;
  if (p.src.degree < 1 || p.src.degree > 3 || p.src.es0 != p.nch) return 1;
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

// Packed render kernel: TWO output pixels per lane (x and x+64 of one row),
// all coordinate arithmetic on float2 so that it issues as v_pk_* instructions
// (see eu_math2.h), divisions/square roots by the range-checked FMA sequences,
// b-spline weights with the structural zeros of the weight matrix skipped.
// Same operations, same order, same bits as eu_render_kernel - only how they
// are issued changes. Covers the jobs without twining whose source is a
// lat/lon image; everything else stays on eu_render_kernel.
#include <cstdlib>
#include "eu_packed_dev.h"

#define EU2_TILE_W 128   // pixels of one row per wave and pass (2 per lane)
#define EU2_TILE_H 4   // default waves (rows) per workgroup
#define EU2_UNIT_ROWS 8  // tile rows per XCD unit

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
  // rotated targets and twined jobs walk their units column by column (eu_xcd_tile): their source lines are
  // shared between vertically neighbouring tiles. EU_HIP_COLMAJOR=0 / 1 forces one walk (A/B runs).
  {
    const char *cme = getenv("EU_HIP_COLMAJOR");          // read on every launch, like the other A/B switches
    const int cm_env = cme && cme[0] ? atoi(cme) : -1;
    const bool cm = cm_env >= 0 ? cm_env != 0 : (p.form != EU_FORM_BA || p.twine);
    if (cm) p.unit_rows = -p.unit_rows;
  }
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

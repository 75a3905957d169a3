// The hot path: one HIP thread per output pixel.
//
//   zimt::process(shape, stepper, environment | twine_t, storer)
//     envutil_payload.cc:541, zimt/wielding.h:151-463
//
// Mapping to CDNA4:
//  * a wavefront owns 64 consecutive x of one output row: the interleaved row
//    write of zimt's storer (put.h:122-136) becomes one contiguous
//    64*NCH*4-byte store per wave;
//  * a 256-thread workgroup owns a 64x4 output tile; workgroup ids are remapped
//    so that each XCD (blockIdx % 8 share one) walks a contiguous band of
//    tiles and its private L2 sees a compact piece of the source;
//  * the steppers' row/segment invariants (stepper.h) are tables in HBM filled
//    by the host once per target: a column table (read coalesced) and a row
//    table (wave-uniform, read through the scalar cache);
//  * source coefficients are gathered straight from the braced container; the
//    (d+1)^2 footprint of neighbouring lanes overlaps, so most taps hit L1/L2;
//  * no MFMA: the work is gather + scalar multiply/add in a fixed order.
//
// Arithmetic contract: every float operation is the one the reference's
// goading back-end performs, in its order, with no FMA contraction (this file
// is compiled with -ffp-contract=off; fp32 division and sqrt are hipcc's
// correctly rounded defaults). See DESIGN.md "Numerics".

#include <hip/hip_runtime.h>
#include <climits>
#include "eu_device.h"
#include "eu_math.h"

#include "eu_render_dev.h"

// ---------------------------------------------------------------------------
// the kernel
// ---------------------------------------------------------------------------

// GEN: the job's stepper is the generic one (EU_FORM_GENERIC: a facet with translation, --single); only the
// run-time-degree variants are instantiated with it
template <int NCH, int DEG, bool TWINE, bool GEN = false>
__global__ __launch_bounds__(256) void eu_render_kernel(const eu_render_params p)
{
  const int b = eu_xcd_tile(blockIdx.x, p.tiles_x, p.tiles_y, -EU_UNIT_ROWS);
  if (b < 0) return;
  const int tile_y = b / p.tiles_x, tile_x = b - tile_y * p.tiles_x;
  const int lane = threadIdx.x & 63;
  const int wrow = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int x = tile_x * EU_TILE_W + lane;
  const int y = p.row_begin + tile_y * EU_TILE_H + wrow;   // wave-uniform
  if (y >= p.row_end || x >= p.width) return;

  const float *rowt = p.row + (long long)eu_frame_row(y, p.band_shift, p.band_count, p.band_index) * EU_ROW_FLOATS;
  const float *col0 = p.col, *col1 = p.col + p.width;
  float rx, ry, rz;
  eu_stepper<GEN>(p, col0, col1, rowt, x, rx, ry, rz);

  float *dst = p.out + (long long)(y - p.row_begin) * p.out_stride;
  if (p.stage == 1) {
    dst[3 * x] = rx; dst[3 * x + 1] = ry; dst[3 * x + 2] = rz;
    return;
  }
  if (p.stage == 2) {
    float sx, sy;
    int face;
    bool hit = eu_source_coordinate(p.src, rx, ry, rz, sx, sy, face);
    bool cube = p.src.prj == EU_CUBEMAP || p.src.prj == EU_BIATAN6;
    dst[3 * x] = hit ? sx : 0.0f;
    dst[3 * x + 1] = hit ? sy : 0.0f;
    dst[3 * x + 2] = hit ? (cube ? (float)face : 0.0f) : -1.0f;
    return;
  }

  if (p.nch_out != NCH) {
    // channel adaption (repix_t): the source has NCH channels, the target nch_out
    const int on = p.nch_out;
    float q4[4] = { 0.0f, 0.0f, 0.0f, 0.0f }, acc[4] = { 0.0f, 0.0f, 0.0f, 0.0f };
    if constexpr (!TWINE) {
      eu_environment_repix<NCH, DEG>(p.src, on, rx, ry, rz, acc);
    } else {
      const float *col2 = p.col + 2 * p.width, *col3 = p.col + 3 * p.width;
      float ax, ay, az, bx, by, bz;
      eu_stepper<GEN>(p, col2, col3, rowt, x, ax, ay, az);
      eu_stepper<GEN>(p, col0, col1, rowt + EU_ROW_VARIANT, x, bx, by, bz);
      float dxx = ax - rx, dxy = ay - ry, dxz = az - rz;
      float dyx = bx - rx, dyy = by - ry, dyz = bz - rz;
      for (int k = 0; k < p.ntaps; k++) {
        float cx = p.taps[3 * k], cy = p.taps[3 * k + 1], cw = p.taps[3 * k + 2];
        eu_environment_repix<NCH, DEG>(p.src, on, rx + cx * dxx + cy * dyx, ry + cx * dxy + cy * dyy,
                                       rz + cx * dxz + cy * dyz, q4);
        for (int c = 0; c < on; c++) acc[c] = acc[c] + cw * q4[c];
      }
    }
    float *o4 = dst + (long long)x * on;
    for (int c = 0; c < on; c++) o4[c] = acc[c];
    return;
  }
  float px[NCH];
  if constexpr (!TWINE) {
    eu_environment<NCH, DEG>(p.src, rx, ry, rz, px);
  } else {
    // deriv_stepper (stepper.h:1591-1715) + twine_t::eval (twining.h:128-263)
    const float *col2 = p.col + 2 * p.width, *col3 = p.col + 3 * p.width;
    float ax, ay, az, bx, by, bz;
    eu_stepper<GEN>(p, col2, col3, rowt, x, ax, ay, az);          // r10: x-biased
    eu_stepper<GEN>(p, col0, col1, rowt + EU_ROW_VARIANT, x, bx, by, bz);      // r01: y-biased
    float dxx = ax - rx, dxy = ay - ry, dxz = az - rz;
    float dyx = bx - rx, dyy = by - ry, dyz = bz - rz;
#pragma unroll
    for (int c = 0; c < NCH; c++) px[c] = 0.0f;
    for (int k = 0; k < p.ntaps; k++) {
      float cx = p.taps[3 * k], cy = p.taps[3 * k + 1], cw = p.taps[3 * k + 2];
      float kx = rx + cx * dxx + cy * dyx;
      float ky = ry + cx * dxy + cy * dyy;
      float kz = rz + cx * dxz + cy * dyz;
      float q[NCH];
      eu_environment<NCH, DEG>(p.src, kx, ky, kz, q);
#pragma unroll
      for (int c = 0; c < NCH; c++) px[c] = px[c] + cw * q[c];
    }
  }
  eu_put<NCH>(dst, x, px);
}

// ---------------------------------------------------------------------------
// the LDS-staged kernel (no twining): the workgroup's 64x4 output tile maps to
// a compact bounding box of source texels. The box is copied into LDS once,
// with coalesced row reads, and the (d+1)^2 taps of every pixel come from
// there - one aligned ds_read_b128 per RGB tap instead of a 12-byte global
// gather through L1. Tiles whose box does not fit (poles of a lat/lon source,
// the +-180 degree seam, strong minification) take the direct path, decided
// per workgroup.
// ---------------------------------------------------------------------------

#define EU_LDS_BYTES (20 * 1024)

__device__ __forceinline__ int eu_wave_min(int v)
{
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = min(v, __shfl_xor(v, o));
  return v;
}
__device__ __forceinline__ int eu_wave_max(int v)
{
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = max(v, __shfl_xor(v, o));
  return v;
}

template <int NCH, int DEG>
__global__ __launch_bounds__(256) void eu_render_lds_kernel(const eu_render_params p)
{
  constexpr int TEX = NCH == 3 ? 4 : NCH;
  constexpr int CAP = EU_LDS_BYTES / (TEX * 4);       // texels
  __shared__ __attribute__((aligned(16))) float tile[CAP * TEX];
  __shared__ int bbw[4][4];

  const int b = eu_xcd_tile(blockIdx.x, p.tiles_x, p.tiles_y, -EU_UNIT_ROWS);
  if (b < 0) return;   // whole workgroup: no barrier is skipped by a part of it
  const int tile_y = b / p.tiles_x, tile_x = b - tile_y * p.tiles_x;
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int x = tile_x * EU_TILE_W + lane;
  const int y = p.row_begin + tile_y * EU_TILE_H + wave;
  const bool active = y < p.row_end && x < p.width;

  float rx = 0.0f, ry = 0.0f, rz = 1.0f;
  if (active) {
    const float *rowt = p.row + (long long)eu_frame_row(y, p.band_shift, p.band_count, p.band_index) * EU_ROW_FLOATS;
    eu_stepper(p, p.col, p.col + p.width, rowt, x, rx, ry, rz);
  }
  float sx = 0.0f, sy = 0.0f;
  int face;
  bool hit = active && eu_source_coordinate(p.src, rx, ry, rz, sx, sy, face);
  int ix = 0, iy = 0;
  float tx = 0.0f, ty = 0.0f;
  if (hit) eu_split<DEG>(p.src, sx, sy, ix, iy, tx, ty);

  // bounding box of the base positions of all hitting lanes
  int mnx = eu_wave_min(hit ? ix : INT_MAX), mny = eu_wave_min(hit ? iy : INT_MAX);
  int mxx = eu_wave_max(hit ? ix : INT_MIN), mxy = eu_wave_max(hit ? iy : INT_MIN);
  if (lane == 0) { bbw[wave][0] = mnx; bbw[wave][1] = mny; bbw[wave][2] = mxx; bbw[wave][3] = mxy; }
  __syncthreads();
  mnx = min(min(bbw[0][0], bbw[1][0]), min(bbw[2][0], bbw[3][0]));
  mny = min(min(bbw[0][1], bbw[1][1]), min(bbw[2][1], bbw[3][1]));
  mxx = max(max(bbw[0][2], bbw[1][2]), max(bbw[2][2], bbw[3][2]));
  mxy = max(max(bbw[0][3], bbw[1][3]), max(bbw[2][3], bbw[3][3]));
  mnx = __builtin_amdgcn_readfirstlane(mnx); mny = __builtin_amdgcn_readfirstlane(mny);
  mxx = __builtin_amdgcn_readfirstlane(mxx); mxy = __builtin_amdgcn_readfirstlane(mxy);
  const bool any = mnx != INT_MAX;
  // window of tap (i, j): base - DEG/2 + {0..DEG}
  const int bx0 = mnx - DEG / 2, by0 = mny - DEG / 2;
  const long long bw = (long long)mxx - mnx + DEG + 1, bh = (long long)mxy - mny + DEG + 1;
  const bool fits = any && bw * bh <= CAP;

  float px[NCH];
#pragma unroll
  for (int c = 0; c < NCH; c++) px[c] = 0.0f;

  if (fits) {
    const int ibw = (int)bw, ibh = (int)bh;
    // stage: wave w copies rows w, w+4, ...; lanes run along x (coalesced)
    for (int r = wave; r < ibh; r += 4) {
      const float *g = p.src.base + (long long)(by0 + r) * p.src.es1 + (long long)bx0 * p.src.es0;
      for (int c = lane; c < ibw; c += 64) {
        const float *q = g + (long long)c * NCH;
        float *d = tile + (r * ibw + c) * TEX;
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
    if (hit) {
      eu_lds_taps<NCH, TEX> lt;
      lt.pitch = ibw * TEX;
      lt.p0 = tile + ((iy - mny) * ibw + (ix - mnx)) * TEX;
      eu_accumulate<NCH, DEG>(p.src.wm, tx, ty, lt, px);
    }
  } else if (hit) {
    eu_global_taps<NCH> g;
    g.es0 = p.src.es0; g.es1 = p.src.es1;
    g.p0 = p.src.base + (long long)(ix - DEG / 2) * p.src.es0 + (long long)(iy - DEG / 2) * p.src.es1;
    eu_accumulate<NCH, DEG>(p.src.wm, tx, ty, g, px);
  }
  if (!active) return;
  if (hit && p.src.brighten != 1.0f) {
    constexpr int ncol = (NCH == 2 || NCH == 4) ? NCH - 1 : NCH;
#pragma unroll
    for (int c = 0; c < ncol; c++) px[c] = px[c] * p.src.brighten;
  }
  eu_put<NCH>(p.out + (long long)(y - p.row_begin) * p.out_stride, x, px);
}

// ---------------------------------------------------------------------------
// launch
// ---------------------------------------------------------------------------

template <int NCH, int DEG>
static hipError_t launch_nd(const eu_render_params &p, hipStream_t st)
{
  dim3 grid((unsigned)eu_xcd_grid(p.tiles_x, p.tiles_y, EU_UNIT_ROWS)), block(256);
  if (p.form == EU_FORM_GENERIC) {
    if constexpr (DEG == -1) {
      if (p.twine) hipLaunchKernelGGL((eu_render_kernel<NCH, -1, true, true>), grid, block, 0, st, p);
      else hipLaunchKernelGGL((eu_render_kernel<NCH, -1, false, true>), grid, block, 0, st, p);
    }
    return hipGetLastError();
  }
  if (p.twine) hipLaunchKernelGGL((eu_render_kernel<NCH, DEG, true>), grid, block, 0, st, p);
  else if (DEG >= 1 && p.stage == 0 && !p.direct && p.nch_out == p.nch) {
    if constexpr (DEG >= 1) hipLaunchKernelGGL((eu_render_lds_kernel<NCH, DEG>), grid, block, 0, st, p);
  }
  else hipLaunchKernelGGL((eu_render_kernel<NCH, DEG, false>), grid, block, 0, st, p);
  return hipGetLastError();
}

template <int NCH>
static hipError_t launch_n(const eu_render_params &p, hipStream_t st)
{
  if (p.form == EU_FORM_GENERIC) return launch_nd<NCH, -1>(p, st);   // generic stepper: run-time-degree kernels only
  switch (p.src.degree) {
    case 0: return launch_nd<NCH, 0>(p, st);
    case 1: return launch_nd<NCH, 1>(p, st);
    case 2: return launch_nd<NCH, 2>(p, st);
    case 3: return launch_nd<NCH, 3>(p, st);
    default: return launch_nd<NCH, -1>(p, st);
  }
}

extern "C" int eu_launch_render(const eu_render_params *pp, void *stream)
{
  eu_render_params p = *pp;
  p.tiles_x = (p.width + EU_TILE_W - 1) / EU_TILE_W;
  p.tiles_y = (p.row_end - p.row_begin + EU_TILE_H - 1) / EU_TILE_H;
  if (p.tiles_x <= 0 || p.tiles_y <= 0) return 0;
  hipStream_t st = (hipStream_t)stream;
  hipError_t e;
  switch (p.nch) {
    case 1: e = launch_n<1>(p, st); break;
    case 2: e = launch_n<2>(p, st); break;
    case 3: e = launch_n<3>(p, st); break;
    case 4: e = launch_n<4>(p, st); break;
    default: return -2;
  }
  return e == hipSuccess ? 0 : -1;
}

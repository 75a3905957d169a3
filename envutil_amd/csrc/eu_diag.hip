// DIAGNOSTIC build of the headline path (NCH 3, degree 3, direct gathers) with
// s_memtime stamps between its phases. Never used by eu_hip_render; its
// timings are read as SHARES of a wave's lifetime, not as kernel time.
#include "eu_render_dev.h"
#include "eu_packed_dev.h"

__device__ __forceinline__ unsigned long long eu_stamp()
{
  unsigned long long t;
  asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t) :: "memory");
  return t;
}

__global__ __launch_bounds__(256) void eu_diag_kernel(const eu_render_params p,
                                                      unsigned long long *stamps)
{
  __builtin_amdgcn_sched_barrier(0);
  unsigned long long t0 = eu_stamp();
  __builtin_amdgcn_sched_barrier(0);
  const int b = eu_xcd_tile(blockIdx.x, p.tiles_x, p.tiles_y, EU_UNIT_ROWS);
  if (b < 0) return;
  const int tile_y = b / p.tiles_x, tile_x = b - tile_y * p.tiles_x;
  const int lane = threadIdx.x & 63;
  const int wrow = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int x = tile_x * EU_TILE_W + lane;
  const int y = p.row_begin + tile_y * EU_TILE_H + wrow;
  if (y >= p.row_end || x >= p.width) return;
  const float *rowt = p.row + (long long)eu_frame_row(y, p.band_shift, p.band_count, p.band_index) * EU_ROW_FLOATS;
  float rx, ry, rz;
  eu_stepper(p, p.col, p.col + p.width, rowt, x, rx, ry, rz);
  asm volatile("" :: "v"(rx), "v"(ry), "v"(rz));
  __builtin_amdgcn_sched_barrier(0);
  unsigned long long t1 = eu_stamp();      // tables loaded, ray formed
  __builtin_amdgcn_sched_barrier(0);
  float sx, sy;
  int face;
  bool hit = eu_source_coordinate(p.src, rx, ry, rz, sx, sy, face);
  int ix, iy;
  float tx, ty;
  eu_split<3>(p.src, sx, sy, ix, iy, tx, ty);
  asm volatile("" :: "v"(ix), "v"(iy), "v"(tx), "v"(ty));
  __builtin_amdgcn_sched_barrier(0);
  unsigned long long t2 = eu_stamp();      // coordinate math done
  __builtin_amdgcn_sched_barrier(0);
  const float *p0 = p.src.base + (long long)(ix - 1) * p.src.es0 + (long long)(iy - 1) * p.src.es1;
  float t[4][4][3];
#pragma unroll
  for (int j = 0; j < 4; j++)
#pragma unroll
    for (int i = 0; i < 4; i++)
#pragma unroll
      for (int c = 0; c < 3; c++) t[j][i][c] = p0[j * p.src.es1 + i * 3 + c];
  __builtin_amdgcn_sched_barrier(0);
  unsigned long long t3 = eu_stamp();      // 16 loads issued
  __builtin_amdgcn_sched_barrier(0);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __builtin_amdgcn_sched_barrier(0);
  unsigned long long t4 = eu_stamp();      // loads landed
  __builtin_amdgcn_sched_barrier(0);
  float wx[4], wy[4], px[3];
  eu_weights<3>(p.src.wm, tx, wx);
  eu_weights<3>(p.src.wm, ty, wy);
#pragma unroll
  for (int c = 0; c < 3; c++) {
    float sum = 0.0f;
#pragma unroll
    for (int j = 0; j < 4; j++) {
      float r = t[j][0][c] * wx[0];
#pragma unroll
      for (int i = 1; i < 4; i++) r = r + wx[i] * t[j][i][c];
      if (j == 0) sum = r * wy[0]; else sum = sum + r * wy[j];
    }
    px[c] = hit ? sum : 0.0f;
  }
  float *o = p.out + (long long)(y - p.row_begin) * p.out_stride + (long long)x * 3;
  o[0] = px[0]; o[1] = px[1]; o[2] = px[2];
  __builtin_amdgcn_sched_barrier(0);
  unsigned long long t5 = eu_stamp();      // accumulate + store issued
  if (lane == 0) {
    unsigned long long *s = stamps + ((long long)blockIdx.x * 4 + wrow) * 8;
    s[0] = t0; s[1] = t1; s[2] = t2; s[3] = t3; s[4] = t4; s[5] = t5;
    s[6] = (unsigned long long)__builtin_amdgcn_s_getreg(((4 - 1) << 11) | (0 << 6) | 20);  // XCC_ID
    s[7] = b;
  }
}

extern "C" int eu_launch_diag(const eu_render_params *pp, unsigned long long *stamps_dev,
                              void *stream)
{
  eu_render_params p = *pp;
  if (p.form == EU_FORM_GENERIC) return -1;     // the phase stamps are for the ordinary steppers
  p.tiles_x = (p.width + EU_TILE_W - 1) / EU_TILE_W;
  p.tiles_y = (p.row_end - p.row_begin + EU_TILE_H - 1) / EU_TILE_H;
  hipLaunchKernelGGL(eu_diag_kernel, dim3((unsigned)eu_xcd_grid(p.tiles_x, p.tiles_y, EU_UNIT_ROWS)), dim3(256), 0,
                     (hipStream_t)stream, p, stamps_dev);
  return hipGetLastError() == hipSuccess ? 0 : -1;
}

// ---------------------------------------------------------------------------
// device self-tests of the range-restricted division / sqrt / packed atan2f
// against hipcc's correctly rounded `/`, sqrtf and the scalar restatement
// ---------------------------------------------------------------------------
#include "eu_math2.h"

__device__ __forceinline__ unsigned long long eu_mix64(unsigned long long &s)
{
  unsigned long long z = (s += 0x9e3779b97f4a7c15ull);
  z = (z ^ (z >> 30)) * 0xbf58476d1ce4e5b9ull;
  z = (z ^ (z >> 27)) * 0x94d049bb133111ebull;
  return z ^ (z >> 31);
}

// a float with a random mantissa and an exponent in [2^-40, 2^40], random sign
__device__ __forceinline__ float eu_rand_safe(unsigned r, int emin, int emax)
{
  unsigned e = 127 + emin + (r >> 23) % (unsigned)(emax - emin + 1);
  return eu_u2f((r & 0x807fffffu) | (e << 23));
}

__global__ void eu_selftest_kernel(unsigned long long seed, int iters, unsigned long long *bad)
{
  __shared__ __attribute__((aligned(16))) float atab[EU_ATAN_TAB_FLOATS];
  if (threadIdx.x < EU_ATAN_TAB_ENTRIES) eu_atan_tab_entry(threadIdx.x, atab + 8 * threadIdx.x);
  __syncthreads();
  unsigned long long s = seed + 0x1000193ull * (blockIdx.x * blockDim.x + threadIdx.x);
  unsigned long long b_div = 0, b_sqrt = 0, b_atan2 = 0, b_cdiv = 0;
  const float c = 6.28318548202514648f, rc = 1.0f / c;
  for (int i = 0; i < iters; i++) {
    unsigned long long r0 = eu_mix64(s), r1 = eu_mix64(s);
    eu_f2 n = { eu_rand_safe((unsigned)r0, -40, 40), eu_rand_safe((unsigned)(r0 >> 32), -40, 40) };
    eu_f2 d = { eu_rand_safe((unsigned)r1, -40, 40), eu_rand_safe((unsigned)(r1 >> 32), -40, 40) };
    if ((i & 7) == 0) { n.x = d.x; n.y = -d.y; }                    // exact quotients
    if ((i & 7) == 1) { n.x = eu_u2f(eu_f2u(d.x) + 1); n.y = 0.0f; } // neighbours, zero numerator
    eu_f2 q = eu_div2_safe(n, d);
    float q0 = n.x / d.x, q1 = n.y / d.y;
    b_div += (eu_f2u(q.x) != eu_f2u(q0)) + (eu_f2u(q.y) != eu_f2u(q1));
    eu_f2 a = eu_abs2(d);
    eu_f2 sq = eu_sqrt2_safe(a);
    b_sqrt += (eu_f2u(sq.x) != eu_f2u(sqrtf(a.x))) + (eu_f2u(sq.y) != eu_f2u(sqrtf(a.y)));
    // ray-like operands (exponents -20..3) and the full safe range alternate
    eu_f2 y = (i & 1) ? n : (eu_f2){ eu_rand_safe((unsigned)r0, -20, 3), eu_rand_safe((unsigned)(r0 >> 32), -20, 3) };
    eu_f2 x = (i & 1) ? d : (eu_f2){ eu_rand_safe((unsigned)r1, -20, 3), eu_rand_safe((unsigned)(r1 >> 32), -20, 3) };
    eu_f2 t = eu_atan2f_2(y, x);
    b_atan2 += (eu_f2u(t.x) != eu_f2u(eu_atan2f(y.x, x.x))) + (eu_f2u(t.y) != eu_f2u(eu_atan2f(y.y, x.y)));
    t = eu_atan2f_2_tab(y, x, atab, 0);
    b_atan2 += (eu_f2u(t.x) != eu_f2u(eu_atan2f(y.x, x.x))) + (eu_f2u(t.y) != eu_f2u(eu_atan2f(y.y, x.y)));
    eu_f2 xp = eu_abs2(x);
    t = eu_atan2f_2_tab(y, xp, atab, 1);
    b_atan2 += (eu_f2u(t.x) != eu_f2u(eu_atan2f(y.x, xp.x))) + (eu_f2u(t.y) != eu_f2u(eu_atan2f(y.y, xp.y)));
    eu_f2 v = { fabsf(eu_rand_safe((unsigned)r0, -30, 3)), fabsf(eu_rand_safe((unsigned)r1, -30, 3)) };
    eu_f2 cd = eu_div2_const(v, c, rc);
    b_cdiv += (eu_f2u(cd.x) != eu_f2u(v.x / c)) + (eu_f2u(cd.y) != eu_f2u(v.y / c));
  }
  if (b_div) atomicAdd(&bad[0], b_div);
  if (b_sqrt) atomicAdd(&bad[1], b_sqrt);
  if (b_atan2) atomicAdd(&bad[2], b_atan2);
  if (b_cdiv) atomicAdd(&bad[3], b_cdiv);
}

extern "C" int eu_launch_selftest(unsigned long long seed, int blocks, int iters,
                                  unsigned long long *bad_dev, void *stream)
{
  hipLaunchKernelGGL(eu_selftest_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, seed,
                     iters, bad_dev);
  return hipGetLastError() == hipSuccess ? 0 : -1;
}


// ---------------------------------------------------------------------------
// the ray -> source coordinate stage on caller-supplied rays (tests of its edge
// cases: exact ties and signed zeros in ray_to_cubeface, axis-aligned and
// denormal rays in the mounts). variant 0: eu_source_coordinate (general kernels),
// 1: eu_coord2 (packed kernels, rays 2k and 2k + 1 in one thread), 2: eu_coord2_ok
// (staged kernel: no fallbacks; a lane that needs one reports face -2).
// out3 = { source x, source y, cube face | 0 }, { 0, 0, -1 } for a miss; the
// packed forms fold the face into y and report 0.
// ---------------------------------------------------------------------------
template <int PRJ>
__device__ void eu_diag_coord_packed(const eu_src_dev &s, const float *rays, long n, long k,
                                     const float *atab, bool only_ok, float *out3)
{
  const long ia = 2 * k, ib = ia + 1 < n ? ia + 1 : ia;
  eu_ray2 r;
  r.x = (eu_f2){ rays[3 * ia], rays[3 * ib] };
  r.y = (eu_f2){ rays[3 * ia + 1], rays[3 * ib + 1] };
  r.z = (eu_f2){ rays[3 * ia + 2], rays[3 * ib + 2] };
  eu_f2 sx, sy;
  eu_i2 hit, ok = { -1, -1 };
  if (only_ok) hit = eu_coord2_ok<PRJ>(s, r, sx, sy, atab, ok);
  else hit = eu_coord2<PRJ>(s, r, sx, sy, atab);
  for (int h = 0; h < 2; h++) {
    const long i = h ? ib : ia;
    if (h && ib == ia) break;
    const bool good = h ? ok.y != 0 : ok.x != 0, hh = h ? hit.y != 0 : hit.x != 0;
    out3[3 * i] = hh ? (h ? sx.y : sx.x) : 0.0f;
    out3[3 * i + 1] = hh ? (h ? sy.y : sy.x) : 0.0f;
    out3[3 * i + 2] = !good ? -2.0f : hh ? 0.0f : -1.0f;
  }
}

__global__ __launch_bounds__(256) void eu_diag_coord_kernel(const eu_src_dev s, const float *rays,
                                                            long n, int variant, float *out3)
{
  __shared__ __attribute__((aligned(16))) float atab[EU_ATAN_TAB_FLOATS];
  if (threadIdx.x < EU_ATAN_TAB_ENTRIES) eu_atan_tab_entry(threadIdx.x, atab + 8 * threadIdx.x);
  __syncthreads();
  const long k = (long)blockIdx.x * 256 + threadIdx.x;
  if (variant == 0) {
    if (k >= n) return;
    float sx, sy;
    int face;
    const bool hit = eu_source_coordinate(s, rays[3 * k], rays[3 * k + 1], rays[3 * k + 2], sx, sy, face);
    out3[3 * k] = hit ? sx : 0.0f;
    out3[3 * k + 1] = hit ? sy : 0.0f;
    out3[3 * k + 2] = hit ? (float)face : -1.0f;
    return;
  }
  if (2 * k >= n) return;
  switch (s.prj) {
    case EU_SPHERICAL: eu_diag_coord_packed<EU_SPHERICAL>(s, rays, n, k, atab, variant == 2, out3); break;
    case EU_CUBEMAP: eu_diag_coord_packed<EU_CUBEMAP>(s, rays, n, k, atab, variant == 2, out3); break;
    case EU_BIATAN6: eu_diag_coord_packed<EU_BIATAN6>(s, rays, n, k, atab, variant == 2, out3); break;
  }
}

extern "C" int eu_launch_diag_coords(const eu_src_dev *s, const float *rays_dev, long n, int variant,
                                     float *out_dev, void *stream)
{
  if (n <= 0) return 0;
  if (variant != 0 && s->prj != EU_SPHERICAL && s->prj != EU_CUBEMAP && s->prj != EU_BIATAN6) return 1;
  hipLaunchKernelGGL(eu_diag_coord_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0,
                     (hipStream_t)stream, *s, rays_dev, n, variant, out_dev);
  return hipGetLastError() == hipSuccess ? 0 : -1;
}

// DIAGNOSTIC build of the headline path (NCH 3, degree 3, direct gathers) with
// s_memtime stamps between its phases. Never used by eu_hip_render; its
// timings are read as SHARES of a wave's lifetime, not as kernel time.
#include "eu_render_dev.h"

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
  const int nblk = p.tiles_x * p.tiles_y;
  int b = eu_xcd_swizzle(blockIdx.x, nblk);
  const int tile_y = b / p.tiles_x, tile_x = b - tile_y * p.tiles_x;
  const int lane = threadIdx.x & 63;
  const int wrow = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int x = tile_x * EU_TILE_W + lane;
  const int y = p.row_begin + tile_y * EU_TILE_H + wrow;
  if (y >= p.row_end || x >= p.width) return;
  const float *rowt = p.row + (long long)y * EU_ROW_FLOATS;
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
  p.tiles_x = (p.width + EU_TILE_W - 1) / EU_TILE_W;
  p.tiles_y = (p.row_end - p.row_begin + EU_TILE_H - 1) / EU_TILE_H;
  hipLaunchKernelGGL(eu_diag_kernel, dim3((unsigned)(p.tiles_x * p.tiles_y)), dim3(256), 0,
                     (hipStream_t)stream, p, stamps_dev);
  return hipGetLastError() == hipSuccess ? 0 : -1;
}

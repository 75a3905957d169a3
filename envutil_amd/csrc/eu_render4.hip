// Packed render kernel with PER-WAVE LDS staging of the b-spline footprint.
//
// A wavefront (a workgroup of its own: no barrier, no tail behind a slower sibling) owns a
// 16x8 output tile (two horizontally adjacent pixels per lane) and a slice of LDS. It
//   1. computes the source coordinates of its 128 pixels (eu_packed_dev.h: the same float2
//      arithmetic as eu_render2_kernel, the reference's operations in the reference's
//      order, without the scalar fallbacks); where the stepper's row constants make the
//      source COLUMN a function of the target column alone (an upright cubemap /
//      rectilinear / cylindrical target of a lat/lon source: stepper.h hoists the same
//      invariants per segment), that half of the chain - the longitude atan2f, the square
//      root, the x gate, split and weights - comes from a per-column table that a small
//      pre-pass kernel fills with the very same device functions,
//   2. reduces the integer base positions to the tile's bounding box (DPP row shifts +
//      row broadcasts, no LDS traffic, no barrier),
//   3. copies the box from the braced container straight into LDS with LDS-DMA
//      (global_load_lds_dwordx4, one instruction per box row: every lane fetches ONE
//      texel, 16 bytes from its own 4-byte-aligned source address - an RGB texel plus one
//      float that is never read - so the LDS image is made of aligned 16-byte texels, rows
//      back to back, without a single VGPR or ds_write),
//   4. evaluates the (d+1)^2 taps of both pixels with ds_read_b128 and the reference's
//      weighted sum (zimt/eval.h:904-1059), and
//   5. stores 24 contiguous bytes per lane (192 per row and wave).
// Every source texel of the box passes the vector memory pipe once per tile instead of
// once per tap and lane quad: the L1 (TCP) tag rate that bounds eu_render2_kernel (7.0
// line accesses per output pixel) drops to the staging traffic (1.8).
//
// Tiles whose box exceeds the LDS slice (the poles of a lat/lon source, the +-180 degree
// seam, strong minification) and tiles with a lane that needs one of the scalar fallbacks
// of the coordinate arithmetic go to a work list and are rendered by the direct-gather
// kernel behind this one.
//
// Stands in for: zimt::process' get/act/put loop (wielding.h:151-463) with the
// evaluator's gathers (zimt/eval.h:838-889) replaced by LDS reads.
#include <cstdlib>
#include <cstdio>
#include <cstring>
#include <vector>
#include <algorithm>
#include "eu_packed_dev.h"

#define EU4_TW 16          // wave tile, pixels
#define EU4_TH 8
#define EU4_WAVES 4        // waves per workgroup of the direct-gather kernel
#ifndef EU4_WAVES0
#define EU4_WAVES0 1       // waves per workgroup of the staged kernel
#define EU4_TEXELS 384     // LDS texels (16 bytes) per wave of the staged kernel: 6 KB + the atanf table (3 KB);
                           // 16 single-wave workgroups per CU (the hardware's limit) use 144 of 160 KB
#define EU4_OCC0 4         // waves per SIMD the registers are capped for
#endif
#define EU4_SHARDS 1024    // work list of the direct-gather kernel: EU4_SHARDS lists, a tile goes to list id % EU4_SHARDS
// layout of eu_render_params::wl (ints); every counter on a 64-byte line of its own
#define EU4_WL_SHARD(s) (16 * (s))                       // entries in list s
#define EU4_WL_DONE1 (16 * EU4_SHARDS)                   // finished workgroups, direct-gather kernel
#define EU4_WL_DYN(x) (16 * (EU4_SHARDS + 1 + (x)))       // eu_render5_kernel: next batch of XCD x's second loop
#define EU4_WL_ENTRIES (16 * (EU4_SHARDS + 16))          // entry k of list s at + k * EU4_SHARDS + s
#define EU4_UNIT_ROWS 4    // tile rows per XCD unit (32 pixel rows)
#ifndef EU4_ASM_TAPS
#define EU4_ASM_TAPS 0      // 1: the tap reads issued by hand (eu4_taps2), 0: left to the compiler
#endif
#define EU4_COL_FLOATS 8   // per-column table: ix, tx, wx[0..3], sqrt(rx^2 + rz^2), longitude
#define EU4_MAX_PLANS 16

// Bounding-box reduction over the wavefront: min of two and max of two registers, all four
// interleaved (a DPP operand written by the preceding VALU instruction needs two wait states;
// three independent instructions sit between dependent ones). Log-step row shifts inside the
// rows of 16 lanes, then the two row broadcasts; lanes without a valid source keep their
// value (the destination is the second operand). Results are valid in lane 63. EXEC must be
// all ones.
__device__ __forceinline__ void eu4_box_reduce(int &mn0, int &mn1, int &mx0, int &mx1)
{
#define EU4_RED(ctrl)                                        \
  "v_min_i32_dpp %0, %0, %0 " ctrl "\n\t"                    \
  "v_min_i32_dpp %1, %1, %1 " ctrl "\n\t"                    \
  "v_max_i32_dpp %2, %2, %2 " ctrl "\n\t"                    \
  "v_max_i32_dpp %3, %3, %3 " ctrl "\n\t"
  asm("s_nop 1\n\t"
      EU4_RED("row_shr:1 row_mask:0xf bank_mask:0xf")
      EU4_RED("row_shr:2 row_mask:0xf bank_mask:0xf")
      EU4_RED("row_shr:4 row_mask:0xf bank_mask:0xf")
      EU4_RED("row_shr:8 row_mask:0xf bank_mask:0xf")
      EU4_RED("row_bcast:15 row_mask:0xa bank_mask:0xf")
      EU4_RED("row_bcast:31 row_mask:0xc bank_mask:0xf")
      "s_nop 0"
      : "+v"(mn0), "+v"(mn1), "+v"(mx0), "+v"(mx1));
#undef EU4_RED
  mn0 = __builtin_amdgcn_readlane(mn0, 63); mn1 = __builtin_amdgcn_readlane(mn1, 63);
  mx0 = __builtin_amdgcn_readlane(mx0, 63); mx1 = __builtin_amdgcn_readlane(mx1, 63);
}

typedef __attribute__((address_space(3))) void *eu4_lds_void;
typedef const __attribute__((address_space(1))) void *eu4_gbl_void;

// The (d+1)^2 taps of BOTH pixels of a lane from the LDS image, window row by window row,
// with the LDS reads issued by hand: the texels of row j + 1 are requested behind the wait for
// row j and ahead of row j's sums, so one row of reads is in flight under one row of arithmetic
// and at most two rows (64 registers) are live. Left to the compiler, all 32 reads of the two
// pixels are hoisted to the top and what they return is spilled to scratch. Asm loads are
// invisible to the compiler's s_waitcnt bookkeeping: every row has its own wait statement that
// names the destinations (lgkmcnt(0): scalar loads may share the counter and return out of
// order). The arithmetic is eu_accumulate1's: the reference's weighted sum in the reference's
// order (zimt/eval.h:904-1059).
typedef float eu4_f4 __attribute__((ext_vector_type(4)));

// `dep`: a value of the arithmetic that has to be finished before these reads are issued (the
// compiler otherwise sinks the sums below all the requests and keeps every row live)
template <int ORDER>
__device__ __forceinline__ void eu4_row_request(unsigned a, unsigned b, eu4_f4 *ta, eu4_f4 *tb, float &dep)
{
  if constexpr (ORDER == 4) {
    asm volatile("ds_read_b128 %0, %8\n\tds_read_b128 %1, %8 offset:16\n\t"
                 "ds_read_b128 %2, %8 offset:32\n\tds_read_b128 %3, %8 offset:48\n\t"
                 "ds_read_b128 %4, %9\n\tds_read_b128 %5, %9 offset:16\n\t"
                 "ds_read_b128 %6, %9 offset:32\n\tds_read_b128 %7, %9 offset:48"
                 : "=&v"(ta[0]), "=&v"(ta[1]), "=&v"(ta[2]), "=&v"(ta[3]),
                   "=&v"(tb[0]), "=&v"(tb[1]), "=&v"(tb[2]), "=&v"(tb[3])
                 : "v"(a), "v"(b), "v"(dep) : "memory");
  } else if constexpr (ORDER == 3) {
    asm volatile("ds_read_b128 %0, %6\n\tds_read_b128 %1, %6 offset:16\n\tds_read_b128 %2, %6 offset:32\n\t"
                 "ds_read_b128 %3, %7\n\tds_read_b128 %4, %7 offset:16\n\tds_read_b128 %5, %7 offset:32"
                 : "=&v"(ta[0]), "=&v"(ta[1]), "=&v"(ta[2]), "=&v"(tb[0]), "=&v"(tb[1]), "=&v"(tb[2])
                 : "v"(a), "v"(b), "v"(dep) : "memory");
  } else {
    asm volatile("ds_read_b128 %0, %4\n\tds_read_b128 %1, %4 offset:16\n\t"
                 "ds_read_b128 %2, %5\n\tds_read_b128 %3, %5 offset:16"
                 : "=&v"(ta[0]), "=&v"(ta[1]), "=&v"(tb[0]), "=&v"(tb[1])
                 : "v"(a), "v"(b), "v"(dep) : "memory");
  }
}

template <int ORDER>
__device__ __forceinline__ void eu4_row_wait(eu4_f4 *ta, eu4_f4 *tb)
{
  if constexpr (ORDER == 4)
    asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(ta[0]), "+v"(ta[1]), "+v"(ta[2]), "+v"(ta[3]),
                                          "+v"(tb[0]), "+v"(tb[1]), "+v"(tb[2]), "+v"(tb[3]) :: "memory");
  else if constexpr (ORDER == 3)
    asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(ta[0]), "+v"(ta[1]), "+v"(ta[2]),
                                          "+v"(tb[0]), "+v"(tb[1]), "+v"(tb[2]) :: "memory");
  else
    asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(ta[0]), "+v"(ta[1]), "+v"(tb[0]), "+v"(tb[1]) :: "memory");
}

// a, b: LDS byte addresses of the two windows; pitch_bytes: bytes per LDS row
template <int NCH, int DEG>
__device__ __forceinline__ void eu4_taps2(unsigned a, unsigned b, unsigned pitch_bytes, const float *wxa,
                                          const float *wya, const float *wxb, const float *wyb,
                                          eu_f2 tx, eu_f2 ty, float *outa, float *outb)
{
  constexpr int order = DEG + 1;
  eu4_f4 ta[2][order], tb[2][order];
  float suma[NCH], sumb[NCH];
  float dep = 0.0f;
  eu4_row_request<order>(a, b, ta[0], tb[0], dep);
#pragma unroll
  for (int j = 0; j < order; j++) {
    eu4_row_wait<order>(ta[j & 1], tb[j & 1]);
    if (j + 1 < order) {
      a += pitch_bytes; b += pitch_bytes;
      // behind the sums of row j - 1
      if (j >= 1) dep = suma[NCH - 1] + sumb[NCH - 1];
      eu4_row_request<order>(a, b, ta[(j + 1) & 1], tb[(j + 1) & 1], dep);
    }
    if constexpr (DEG == 1) {
      // _eval_linear, eval.h:1014-1059: wl = 1 - t, wr = t
      const float wl0a = 1.0f - tx.x, wr0a = tx.x, wl0b = 1.0f - tx.y, wr0b = tx.y;
#pragma unroll
      for (int c = 0; c < NCH; c++) {
        float ra = ta[j & 1][0][c] * wl0a; ra = ra + ta[j & 1][1][c] * wr0a;
        float rb = tb[j & 1][0][c] * wl0b; rb = rb + tb[j & 1][1][c] * wr0b;
        if (j == 0) { suma[c] = ra * (1.0f - ty.x); sumb[c] = rb * (1.0f - ty.y); }
        else { suma[c] = suma[c] + ra * ty.x; sumb[c] = sumb[c] + rb * ty.y; }
      }
    } else {
#pragma unroll
      for (int c = 0; c < NCH; c++) {
        float ra = ta[j & 1][0][c] * wxa[0], rb = tb[j & 1][0][c] * wxb[0];
#pragma unroll
        for (int i = 1; i < order; i++) { ra = ra + wxa[i] * ta[j & 1][i][c]; rb = rb + wxb[i] * tb[j & 1][i][c]; }
        if (j == 0) { suma[c] = ra * wya[0]; sumb[c] = rb * wyb[0]; }
        else { suma[c] = suma[c] + ra * wya[j]; sumb[c] = sumb[c] + rb * wyb[j]; }
      }
    }
  }
#pragma unroll
  for (int c = 0; c < NCH; c++) { outa[c] = suma[c]; outb[c] = sumb[c]; }
}



// one LDS-DMA row: lane c fetches 16 bytes at sb + voff into LDS dst + 16 c (M0 = dst is written
// in the statement that uses it and restored for the compiler)
__device__ __forceinline__ void eu4_dma_row(unsigned dst, unsigned voff, const char *sb)
{
  unsigned keep;
  asm volatile("s_mov_b32 %0, m0\n\t"
               "s_mov_b32 m0, %1\n\t"
               "s_nop 0\n\t"
               "global_load_lds_dwordx4 %2, %3\n\t"
               "s_mov_b32 m0, %0"
               : "=&s"(keep) : "s"(dst), "v"(voff), "s"(sb) : "memory");
}

// what the staged kernel needs besides eu_render_params
struct eu4_plan {
  const float *atab_g;    // the atanf range table (eu_math2.h) in global memory, 768 floats
  const int *tileplan;    // per tile row of the launch: column table to use, -1: none
  const float *coltab;    // [plan][width][EU4_COL_FLOATS]
  int tiles16;            // wave tiles per tile row
  // eu_render5_kernel's second loop (the tile rows without a common column plan), dealt out dynamically: the tile
  // rows XCD x owns in that loop are l2_rows[l2_off[x] .. l2_off[x + 1]) (raster order), a dequeue is one BATCH of
  // two neighbouring tiles of a row; l2_half = ceil(tiles16 / 2) batches per row, l2_magic = floor(2^40 / l2_half) + 1
  const int *l2_rows;
  int l2_off[9];
  int l2_half;
  unsigned long long l2_magic;
  // the first loop's list, the same way: {double row m, its column plan} pairs of XCD x at l1_rows[2 * l1_off[x] ...],
  // walked with a fixed stride (equal tiles: no queue); l1_magic = floor(2^40 / tiles16) + 1
  const int *l1_rows;
  int l1_off[9];
  unsigned long long l1_magic;
#ifdef EU5_STAMPS
  unsigned long long *stamps;   // diagnostic build: 8 s_memtime stamps per tile of eu_render5_kernel
#endif
};

#ifdef EU5_STAMPS
#define EU5_STAMP(k) do { asm volatile("" ::: "memory"); st_[k] = __builtin_amdgcn_s_memtime(); asm volatile("" ::: "memory"); } while (0)
#else
#define EU5_STAMP(k) do { } while (0)
#endif

// ---------------------------------------------------------------------------
// one tile of the staged kernel. HOIST: the x half of the coordinate chain comes from the
// column table `ct` (this tile row's plan).
// ---------------------------------------------------------------------------
template <int NCH, int DEG, int PRJ, bool HOIST>
__device__ __forceinline__ void eu4_tile(const eu_render_params &p, const eu4_plan &w, const float *atab,
                                         float *wtile, const float *ct, int tile_y, int x0, int lane)
{
  constexpr int TEX = 4;                              // floats per LDS texel
  constexpr int order = DEG + 1;
  const eu_src_dev &s = p.src;
  const int lx = lane & 7, ly = lane >> 3;
  const int y = p.row_begin + tile_y * EU4_TH + ly;
  const bool yin = y < p.row_end;
  const int yc = yin ? y : p.row_end - 1;
  // the stepper's row constants of this lane's row (stepper.h: per-segment invariants)
  const float *rt = p.row + (long long)eu_frame_row(yc, p.band_shift, p.band_count, p.band_index) * EU_ROW_FLOATS;
  const int xa = x0 + 2 * lx, xb = xa + 1;
  const bool va = yin && xa < p.width, vb = yin && xb < p.width;
  const int xac = xa < p.width ? xa : p.width - 1, xbc = xb < p.width ? xb : p.width - 1;

  eu_f2 tx, ty, gy;
  eu_i2 hit, ok = { -1, -1 };
  int ixa, ixb;
  float wxa[order], wxb[order];
  if constexpr (HOIST) {
    // column entries of the two pixels: ix, tx, wx[0..3], sqrt(rx^2 + rz^2), longitude
    const float4 *ea = (const float4 *)(ct + (size_t)xac * EU4_COL_FLOATS);
    const float4 *eb = (const float4 *)(ct + (size_t)xbc * EU4_COL_FLOATS);
    const float4 a0 = ea[0], a1 = ea[1], b0 = eb[0], b1 = eb[1];
    const float A1 = rt[1], B1 = rt[4];
    const eu_f2 c0 = { p.col[xac], p.col[xbc] };
    const eu_f2 ryy = B1 * c0 + A1;
    // the atanf table is complete behind this wait (LDS-DMA, issued at kernel entry)
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    ixa = __float_as_int(a0.x); ixb = __float_as_int(b0.x);
    ok = ok & (eu_i2){ ixa != INT_MIN ? -1 : 0, ixb != INT_MIN ? -1 : 0 };
    tx = (eu_f2){ a0.y, b0.y };
    if constexpr (DEG >= 2) {
      wxa[0] = a0.z; wxa[1] = a0.w; wxa[2] = a1.x; wxb[0] = b0.z; wxb[1] = b0.w; wxb[2] = b1.x;
      if constexpr (DEG == 3) { wxa[3] = a1.y; wxb[3] = b1.y; }
    } else {
#pragma unroll
      for (int i = 0; i < order; i++) { wxa[i] = 0.0f; wxb[i] = 0.0f; }
    }
    const eu_f2 qs = { a1.z, b1.z };
    // ray_to_ll_t's latitude (geometry.h:297-299) and the y half of md_to_spline
    // (environment.h:988-1006): eu_coord2_ok's operations on the y coordinate
    const eu_f2 lat = eu_atan2f_2_tab_ok(ryy, qs, atab, 1, ok);
    hit = (eu_i2){ -1, -1 };
    if (!s.always_hit) {
      const eu_f2 lon = { a1.w, b1.w };
      hit = (lon >= s.wex0) & (lon <= s.wex1) & (lat >= s.wex2) & (lat <= s.wex3);
    }
    eu_f2 i1 = { (float)((double)lat.x - s.tex_y0), (float)((double)lat.y - s.tex_y0) };
    if (s.cdiv_ok) i1 = eu_div2_const(i1, s.ext_h, s.rcp_ext_h);
    else i1 = i1 / s.ext_h;
    i1 = i1 * s.total_h; i1 = i1 - .5f;
    const eu_f2 sy = i1 - s.win_y_off;
    gy = eu_gate2_ok(sy, s.gate1, s.lower1, s.upper1, ok);
  } else {
    // rays of both pixels (stepper.h: ray = B * c0 (+ C * c1) + A)
    eu_ray2 r;
    {
      const float A0 = rt[0], A1 = rt[1], A2 = rt[2], B0 = rt[3], B1 = rt[4], B2 = rt[5];
      const eu_f2 c0 = { p.col[xac], p.col[xbc] };
      if (p.form == EU_FORM_BCA) {
        const float C0 = rt[6], C1 = rt[7], C2 = rt[8];
        const float *colB = p.col + p.width;
        const eu_f2 c1 = { colB[xac], colB[xbc] };
        r.x = B0 * c0 + C0 * c1 + A0;
        r.y = B1 * c0 + C1 * c1 + A1;
        r.z = B2 * c0 + C2 * c1 + A2;
      } else {
        r.x = B0 * c0 + A0;
        r.y = B1 * c0 + A1;
        r.z = B2 * c0 + A2;
      }
      if (p.norm_mode == EU_NORM_DIV) {
        eu_f2 sqn = r.x * r.x; sqn = sqn + r.y * r.y; sqn = sqn + r.z * r.z;
        const eu_f2 n = { sqrtf(sqn.x), sqrtf(sqn.y) };
        r.x = r.x / n; r.y = r.y / n; r.z = r.z / n;
      }
    }
    // the atanf table is complete behind this wait (LDS-DMA, issued at kernel entry)
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    eu_f2 sx, sy;
    hit = eu_coord2_ok<PRJ>(s, r, sx, sy, atab, ok);
    const eu_f2 gx = eu_gate2_ok(sx, s.gate0, s.lower0, s.upper0, ok);
    gy = eu_gate2_ok(sy, s.gate1, s.lower1, s.upper1, ok);
    eu_f2 fx;
    if constexpr (DEG & 1) fx = (eu_f2){ floorf(gx.x), floorf(gx.y) };
    else fx = (eu_f2){ roundf(gx.x), roundf(gx.y) };
    tx = gx - fx;
    ixa = (int)fx.x; ixb = (int)fx.y;
  }
  hit = hit & (eu_i2){ va ? -1 : 0, vb ? -1 : 0 };
  eu_f2 fy;
  if constexpr (DEG & 1) fy = (eu_f2){ floorf(gy.x), floorf(gy.y) };
  else fy = (eu_f2){ roundf(gy.x), roundf(gy.y) };
  ty = gy - fy;
  const int iya = (int)fy.x, iyb = (int)fy.y;

  // bounding box of the base positions of the tile's hitting pixels
  int mnx = INT_MAX, mny = INT_MAX, mxx = INT_MIN, mxy = INT_MIN;
  if (hit.x) { mnx = ixa; mxx = ixa; mny = iya; mxy = iya; }
  if (hit.y) { mnx = min(mnx, ixb); mxx = max(mxx, ixb); mny = min(mny, iyb); mxy = max(mxy, iyb); }
  eu4_box_reduce(mnx, mny, mxx, mxy);
  const bool any = mnx != INT_MAX;
  const long long bw = (long long)mxx - mnx + order, bh = (long long)mxy - mny + order;
  // the LDS image is the box itself, rows of bw texels back to back (pitch = bw: no padding;
  // rows an odd number of 16-byte texels apart spread over the banks)
  const bool fits = bw <= 64 && bh <= EU4_TEXELS && bw * bh <= EU4_TEXELS;
  // a hitting pixel that left the fast path of the coordinate arithmetic?
  const bool clean = __ballot((hit.x && !ok.x) || (hit.y && !ok.y)) == 0ull;
  float *const orow = p.out + (long long)(yc - p.row_begin) * p.out_stride;
  if (any && !(fits && clean)) {
    // left to the direct-gather kernel: one of EU4_SHARDS lists (a single counter serialises:
    // ~88 returning atomics per microsecond on one word)
    if (lane == 0) {
      const int id = tile_y * w.tiles16 + x0 / EU4_TW;
      const int sh = (int)(((unsigned)id * 0x9E3779B1u) >> 22) & (EU4_SHARDS - 1);   // eu4_shard_of
      const int slot = atomicAdd(p.wl + EU4_WL_SHARD(sh), 1);
      p.wl[EU4_WL_ENTRIES + (size_t)slot * EU4_SHARDS + sh] = id;
    }
    return;
  }
  float qa[NCH], qb[NCH];
  if (any) {
    // stage: one LDS-DMA instruction per box row, lane c fetches the texel of box column c;
    // the address arithmetic is scalar
    const int ibw = (int)bw, ibh = (int)bh;
    const unsigned lds_tile = (unsigned)(unsigned long long)(eu4_lds_void)wtile;
    if (lane < ibw) {
      const int bx0 = mnx - DEG / 2, by0 = mny - DEG / 2;
      const unsigned voff = (unsigned)(lane * NCH) * 4u;
      const char *sb = (const char *)(s.base + ((long long)by0 * s.es1 + (long long)bx0 * NCH));
      const long long step = s.es1 * 4;
      unsigned dst = lds_tile;
      const unsigned dstep = (unsigned)ibw * (TEX * 4u);
#pragma unroll 1
      for (int it = 0; it < ibh; it++) { eu4_dma_row(dst, voff, sb); sb += step; dst += dstep; }
    }
    // the weights, behind the DMA issue
    eu_f2 wx[order], wy[order];
    float wya[order], wyb[order];
    if constexpr (DEG >= 2) {
      eu_weights2<DEG>(s.wm, ty, wy);
      if constexpr (!HOIST) eu_weights2<DEG>(s.wm, tx, wx);
    }
#pragma unroll
    for (int i = 0; i < order; i++) {
      if constexpr (DEG >= 2) {
        wya[i] = wy[i].x; wyb[i] = wy[i].y;
        if constexpr (!HOIST) { wxa[i] = wx[i].x; wxb[i] = wx[i].y; }
      } else {
        wya[i] = wyb[i] = 0.0f;
        if constexpr (!HOIST) { wxa[i] = wxb[i] = 0.0f; }
      }
    }
    // lanes without a hit read the box origin
    const int oa = hit.x ? ((iya - mny) * ibw + (ixa - mnx)) * TEX : 0;
    const int ob = hit.y ? ((iyb - mny) * ibw + (ixb - mnx)) * TEX : 0;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#if EU4_ASM_TAPS
    eu4_taps2<NCH, DEG>(lds_tile + (unsigned)oa * 4u, lds_tile + (unsigned)ob * 4u, (unsigned)ibw * (TEX * 4u),
                        wxa, wya, wxb, wyb, tx, ty, qa, qb);
#else
    {
      eu_lptr lt = (eu_lptr)wtile;
      eu_accumulate1<NCH, DEG, TEX, int, eu_lptr>(lt + oa, ibw * TEX, wxa, wya, tx.x, ty.x, qa);
      eu_accumulate1<NCH, DEG, TEX, int, eu_lptr>(lt + ob, ibw * TEX, wxb, wyb, tx.y, ty.y, qb);
    }
#endif
  } else {
#pragma unroll
    for (int c = 0; c < NCH; c++) { qa[c] = 0.0f; qb[c] = 0.0f; }
  }
  // environment::eval brighten (environment.h:1821-1842), zero on a miss; storer
  constexpr int ncol = (NCH == 2 || NCH == 4) ? NCH - 1 : NCH;
  const bool bright = s.brighten != 1.0f;
#pragma unroll
  for (int c = 0; c < NCH; c++) {
    float a = qa[c], bb = qb[c];
    if (bright && c < ncol) { a = a * s.brighten; bb = bb * s.brighten; }
    qa[c] = hit.x ? a : 0.0f;
    qb[c] = hit.y ? bb : 0.0f;
  }
  if (va) eu_put<NCH>(orow, xa, qa);
  if (vb) eu_put<NCH>(orow, xb, qb);
}

// the staged kernel: one 16x8 tile per wave, EU4_WAVES0 waves (neighbouring tiles of one tile
// row) per workgroup; they share nothing but the atanf table
template <int NCH, int DEG, int PRJ>
__global__ __launch_bounds__(64 * EU4_WAVES0, EU4_OCC0) void eu_render4s_kernel(const eu_render_params p, const eu4_plan w)
{
  __shared__ __attribute__((aligned(16))) float tile_all[EU4_WAVES0 * EU4_TEXELS * 4];
  __shared__ __attribute__((aligned(16))) float atab[768];   // EU_ATAN_TAB_FLOATS, rounded up to three DMA pieces
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  float *const tile = tile_all + wave * (EU4_TEXELS * 4);
  if (PRJ != EU_CUBEMAP && wave == 0) {
    // the table comes from its copy in global memory by LDS-DMA: 3 instructions of 64 x 16
    // bytes cover its 2592 bytes (a workgroup of one wave would spend ~70 VALU instructions
    // computing it)
    static_assert(EU_ATAN_TAB_FLOATS * 4 <= 3 * 1024, "three pieces");
#pragma unroll
    for (int i = 0; i < 3; i++)
      __builtin_amdgcn_global_load_lds((eu4_gbl_void)(w.atab_g + i * 256 + lane * 4),
                                       (eu4_lds_void)(atab + i * 256), 16, 0, 0);
  }
  // XCD-aware tile order without a division: grid = (8 * tiles16, unit rows, rounds).
  // Workgroups are dispatched x fastest and dealt round-robin to the 8 XCDs, so
  // blockIdx.x & 7 is the XCD (up to a rotation); an XCD walks the tiles of its unit
  // (EU4_UNIT_ROWS tile rows) in raster order and gets every 8th unit of the frame.
  const int tile_x = (int)(blockIdx.x >> 3) * EU4_WAVES0 + wave;
  const int tile_y = ((int)blockIdx.z * 8 + (int)(blockIdx.x & 7)) * EU4_UNIT_ROWS + (int)blockIdx.y;
  if (tile_y >= p.tiles_y) return;                    // the whole workgroup
  if constexpr (EU4_WAVES0 > 1) {
    // wave 0's table has to be in LDS before the others use it
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
  }
  if (tile_x >= w.tiles16) return;
  const int plan = PRJ == EU_SPHERICAL ? w.tileplan[tile_y] : -1;
  if (plan >= 0)
    eu4_tile<NCH, DEG, PRJ, PRJ == EU_SPHERICAL>(p, w, atab, tile, w.coltab + (size_t)plan * p.width * EU4_COL_FLOATS,
                                                  tile_y, tile_x * EU4_TW, lane);
  else
    eu4_tile<NCH, DEG, PRJ, false>(p, w, atab, tile, nullptr, tile_y, tile_x * EU4_TW, lane);
}

// one 16x8 tile by direct gathers: (d+1)^2 taps from global memory, every scalar fallback of the
// coordinate arithmetic (the work-list kernel's tile)
template <int NCH, int DEG, int PRJ>
__device__ __forceinline__ void eu4_direct_tile(const eu_render_params &p, const float *atab, int tile_y, int x0, int lane)
{
  const eu_src_dev &s = p.src;
  const int lx = lane & 7, ly = lane >> 3;
  const int y = p.row_begin + tile_y * EU4_TH + ly;
  const bool yin = y < p.row_end;
  const int yc = yin ? y : p.row_end - 1;
  const float *rt = p.row + (long long)eu_frame_row(yc, p.band_shift, p.band_count, p.band_index) * EU_ROW_FLOATS;
  const int xa = x0 + 2 * lx, xb = xa + 1;
  const bool va = yin && xa < p.width, vb = yin && xb < p.width;
  const int xac = xa < p.width ? xa : p.width - 1, xbc = xb < p.width ? xb : p.width - 1;
  eu_ray2 ry;
  {
    const float A0 = rt[0], A1 = rt[1], A2 = rt[2], B0 = rt[3], B1 = rt[4], B2 = rt[5];
    const eu_f2 c0 = { p.col[xac], p.col[xbc] };
    if (p.form == EU_FORM_BCA) {
      const float C0 = rt[6], C1 = rt[7], C2 = rt[8];
      const float *colB = p.col + p.width;
      const eu_f2 c1 = { colB[xac], colB[xbc] };
      ry.x = B0 * c0 + C0 * c1 + A0;
      ry.y = B1 * c0 + C1 * c1 + A1;
      ry.z = B2 * c0 + C2 * c1 + A2;
    } else {
      ry.x = B0 * c0 + A0;
      ry.y = B1 * c0 + A1;
      ry.z = B2 * c0 + A2;
    }
    if (p.norm_mode == EU_NORM_DIV) {
      eu_f2 sqn = ry.x * ry.x; sqn = sqn + ry.y * ry.y; sqn = sqn + ry.z * ry.z;
      const eu_f2 n = { sqrtf(sqn.x), sqrtf(sqn.y) };
      ry.x = ry.x / n; ry.y = ry.y / n; ry.z = ry.z / n;
    }
  }
  eu_f2 sx, sy;
  eu_i2 hit = eu_coord2<PRJ>(s, ry, sx, sy, atab);
  hit = hit & (eu_i2){ va ? -1 : 0, vb ? -1 : 0 };
  float pxa[NCH], pxb[NCH];
  eu_eval2<NCH, DEG>(s, sx, sy, hit, pxa, pxb);
  float *const orow = p.out + (long long)(yc - p.row_begin) * p.out_stride;
  if (va) eu_put<NCH>(orow, xa, pxa);
  if (vb) eu_put<NCH>(orow, xb, pxb);
}

#include "eu_render5.h"

// ---------------------------------------------------------------------------
// pre-pass: the column table of one plan. Thread t fills columns 2t and 2t + 1 with the
// operations the staged kernel performs on the x coordinate (eu_coord2_ok for a lat/lon
// source, the x gate, split and weights), on the same float2 functions: same bits.
// ---------------------------------------------------------------------------
template <int DEG>
__global__ __launch_bounds__(256) void eu_colplan_kernel(const eu_render_params p, float *ct,
                                                         float A0, float A2, float B0, float B2)
{
  __shared__ __attribute__((aligned(16))) float atab[EU_ATAN_TAB_FLOATS];
  if (threadIdx.x < EU_ATAN_TAB_ENTRIES) eu_atan_tab_entry(threadIdx.x, atab + 8 * threadIdx.x);
  __syncthreads();
  constexpr int order = DEG + 1;
  const eu_src_dev &s = p.src;
  const int xa = 2 * (blockIdx.x * 256 + threadIdx.x), xb = xa + 1;
  if (xa >= p.width) return;
  const int xbc = xb < p.width ? xb : p.width - 1;
  const eu_f2 c0 = { p.col[xa], p.col[xbc] };
  const eu_f2 rx = B0 * c0 + A0, rz = B2 * c0 + A2;
  eu_i2 ok = { -1, -1 };
  const eu_f2 q2 = rx * rx + rz * rz;
  const eu_f2 qs = eu_sqrt2_ok(q2, ok);
  const eu_f2 lon = eu_atan2f_2_tab_ok(rx, rz, atab, 0, ok);
  eu_f2 i0 = { (float)((double)lon.x - s.tex_x0), (float)((double)lon.y - s.tex_x0) };
  if (s.cdiv_ok) i0 = eu_div2_const(i0, s.ext_w, s.rcp_ext_w);
  else i0 = i0 / s.ext_w;
  i0 = i0 * s.total_w; i0 = i0 - .5f;
  const eu_f2 sx = i0 - s.win_x_off;
  const eu_f2 gx = eu_gate2_ok(sx, s.gate0, s.lower0, s.upper0, ok);
  eu_f2 fx;
  if constexpr (DEG & 1) fx = (eu_f2){ floorf(gx.x), floorf(gx.y) };
  else fx = (eu_f2){ roundf(gx.x), roundf(gx.y) };
  const eu_f2 tx = gx - fx;
  eu_f2 wx[4];
#pragma unroll
  for (int i = 0; i < 4; i++) wx[i] = (eu_f2){ 0.0f, 0.0f };
  if constexpr (DEG >= 2) eu_weights2<DEG>(s.wm, tx, wx);
  float e[2][EU4_COL_FLOATS];
  e[0][0] = __int_as_float(ok.x ? (int)fx.x : INT_MIN);
  e[1][0] = __int_as_float(ok.y ? (int)fx.y : INT_MIN);
  e[0][1] = tx.x; e[1][1] = tx.y;
#pragma unroll
  for (int i = 0; i < 4; i++) { e[0][2 + i] = i < order ? wx[i].x : 0.0f; e[1][2 + i] = i < order ? wx[i].y : 0.0f; }
  e[0][6] = qs.x; e[1][6] = qs.y;
  e[0][7] = lon.x; e[1][7] = lon.y;
  float *o = ct + (size_t)xa * EU4_COL_FLOATS;
#pragma unroll
  for (int i = 0; i < EU4_COL_FLOATS; i++) o[i] = e[0][i];
  if (xb < p.width) {
#pragma unroll
    for (int i = 0; i < EU4_COL_FLOATS; i++) o[EU4_COL_FLOATS + i] = e[1][i];
  }
}

// ---------------------------------------------------------------------------
// the direct-gather kernel: the tiles of the work list, (d+1)^2 taps from global memory,
// every scalar fallback of the coordinate arithmetic; the last workgroup to finish
// empties the list for the next launch pair
// ---------------------------------------------------------------------------
template <int NCH, int DEG, int PRJ>
__global__ __launch_bounds__(256, 4) void eu_render4d_kernel(const eu_render_params p, const eu4_plan w)
{
  __shared__ __attribute__((aligned(16))) float atab[EU_ATAN_TAB_FLOATS];
  if constexpr (PRJ != EU_CUBEMAP) {
    if (threadIdx.x < EU_ATAN_TAB_ENTRIES) eu_atan_tab_entry(threadIdx.x, atab + 8 * threadIdx.x);
  }
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  // this wave's list: waves gid, gid + EU4_SHARDS, ... share list gid % EU4_SHARDS
  const int gid = blockIdx.x * EU4_WAVES + wave;
  const int sh = gid & (EU4_SHARDS - 1);
  const int nwork = __builtin_amdgcn_readfirstlane(
    __hip_atomic_load(p.wl + EU4_WL_SHARD(sh), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
  __syncthreads();
#pragma unroll 1
  for (int k = gid / EU4_SHARDS; k < nwork; k += (int)(gridDim.x * EU4_WAVES / EU4_SHARDS)) {
    const int id = __builtin_amdgcn_readfirstlane(p.wl[EU4_WL_ENTRIES + (size_t)k * EU4_SHARDS + sh]);
    const int tile_y = id / w.tiles16;
    const int x0 = (id - tile_y * w.tiles16) * EU4_TW;
    eu4_direct_tile<NCH, DEG, PRJ>(p, atab, tile_y, x0, lane);
  }
  __shared__ int last;
  __syncthreads();
  if (threadIdx.x == 0) last = atomicAdd(p.wl + EU4_WL_DONE1, 1) == (int)gridDim.x - 1;
  __syncthreads();
  if (last) {
    for (int i = threadIdx.x; i <= EU4_SHARDS + 8; i += 256)         // the lists, the counter of this kernel, the staged kernel's queues
      __hip_atomic_store(p.wl + 16 * i, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
}

#ifndef EU4_DIRECT_WGS
#define EU4_DIRECT_WGS 2048
#endif

template <int NCH, int DEG, int PRJ>
static int launch4_ndp(const eu_render_params &p, const eu4_plan &w, hipStream_t st)
{
  // EU_HIP_R5: 1 (default) the persistent staged kernel of round 3, 0 round 2's one-tile-per-workgroup form
  // (unset: the persistent kernel for lat/lon sources - headline 1.13 vs 1.22 ms for the launch-level hybrid of
  // the direct-gather kernels -, round 2's form for cubemap sources: config 3 1.00 vs 1.22 ms)
  const char *r5env = getenv("EU_HIP_R5");
  const bool use5 = r5env ? r5env[0] != '0' : PRJ == EU_SPHERICAL;
  if (use5) {
    // as many workgroups as are resident at once (a persistent kernel must not queue a second round)
    static const int per_cu_env = [] { const char *e = getenv("EU_HIP_R5_WGS"); return e ? atoi(e) : 0; }();   // A/B runs
    static const int cus = [] {
      int dev = 0, n = 0;
      if (hipGetDevice(&dev) != hipSuccess) return 0;
      if (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) return 0;
      return n;
    }();
    // eu_div2_rr's range: the divisor in [2^-20, 2^20], the extent's origin 0 or in that range (so that a
    // non-zero difference 'angle - origin' is at least 2^-73)
    auto mag_ok = [](double v) { const double a = v < 0 ? -v : v; return a == 0.0 || (a >= 0x1p-20 && a <= 0x1p20); };
    const bool fast = p.form == EU_FORM_BA && p.norm_mode == EU_NORM_NONE && p.band_count <= 1 && p.src.brighten == 1.0f &&
                      (p.src.prj != EU_SPHERICAL ||
                       (p.src.always_hit && mag_ok(p.src.tex_x0) && mag_ok(p.src.tex_y0) && p.src.ext_w >= 0x1p-20f &&
                        p.src.ext_w <= 0x1p20f && p.src.ext_h >= 0x1p-20f && p.src.ext_h <= 0x1p20f && p.tab_finite));
    static const bool dbg = getenv("EU_HIP_DEBUG") != nullptr;
    if (dbg) fprintf(stderr, "eu_render5: fast %d (form %d norm %d bands %d brighten %g always_hit %d cdiv_ok %d)\n", (int)fast, p.form,
                     p.norm_mode, p.band_count, (double)p.src.brighten, p.src.always_hit, p.src.cdiv_ok);
    int per_cu = 0;
    if (fast) {
      static const int occ = [] { int n = 0; return hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, eu_render5_kernel<NCH, DEG, PRJ, true>, 64 * EU5_WAVES, 0) == hipSuccess ? n : 0; }();
      per_cu = occ;
    } else {
      static const int occ = [] { int n = 0; return hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, eu_render5_kernel<NCH, DEG, PRJ, false>, 64 * EU5_WAVES, 0) == hipSuccess ? n : 0; }();
      per_cu = occ;
    }
    if (per_cu_env > 0) per_cu = std::min(per_cu, per_cu_env);
    const int wgs = (cus / 8) * 8 * per_cu;
    if (wgs <= 0) return -1;
    if (fast) hipLaunchKernelGGL((eu_render5_kernel<NCH, DEG, PRJ, true>), dim3((unsigned)wgs), dim3(64 * EU5_WAVES), 0, st, p, w);
    else hipLaunchKernelGGL((eu_render5_kernel<NCH, DEG, PRJ, false>), dim3((unsigned)wgs), dim3(64 * EU5_WAVES), 0, st, p, w);
  } else {
    const int units = (p.tiles_y + EU4_UNIT_ROWS - 1) / EU4_UNIT_ROWS;
    dim3 grid((unsigned)(8 * ((w.tiles16 + EU4_WAVES0 - 1) / EU4_WAVES0)), (unsigned)EU4_UNIT_ROWS, (unsigned)((units + 7) / 8));
    hipLaunchKernelGGL((eu_render4s_kernel<NCH, DEG, PRJ>), grid, dim3(64 * EU4_WAVES0), 0, st, p, w);
  }
  if (hipGetLastError() != hipSuccess) return -1;
  hipLaunchKernelGGL((eu_render4d_kernel<NCH, DEG, PRJ>), dim3(EU4_DIRECT_WGS), dim3(256), 0, st, p, w);
  if (hipGetLastError() != hipSuccess) {
    // the staged kernel has filled the lists and nobody will empty them: the next job must not append to them
    (void)hipMemsetAsync(p.wl, 0, EU4_WL_ENTRIES * sizeof(int), st);
    return -1;
  }
  return 0;
}

template <int NCH, int DEG>
static int launch4_nd(const eu_render_params &p, const eu4_plan &w, hipStream_t st)
{
  switch (p.src.prj) {
    case EU_SPHERICAL: return launch4_ndp<NCH, DEG, EU_SPHERICAL>(p, w, st);
    case EU_CUBEMAP: return launch4_ndp<NCH, DEG, EU_CUBEMAP>(p, w, st);
    case EU_BIATAN6: return launch4_ndp<NCH, DEG, EU_BIATAN6>(p, w, st);
  }
  return 1;
}

template <int NCH>
static int launch4_n(const eu_render_params &p, const eu4_plan &w, hipStream_t st)
{
  switch (p.src.degree) {
    case 1: return launch4_nd<NCH, 1>(p, w, st);
    case 2: return launch4_nd<NCH, 2>(p, w, st);
    case 3: return launch4_nd<NCH, 3>(p, w, st);
  }
  return 1;
}

// ints the work list buffer needs for a launch of `ntiles` wave tiles
extern "C" size_t eu_render4_worklist_ints(size_t ntiles)
{
  return EU4_WL_ENTRIES + ((ntiles + EU4_SHARDS - 1) / EU4_SHARDS) * EU4_SHARDS;
}
extern "C" size_t eu_render4_worklist_header_ints(void) { return EU4_WL_ENTRIES; }

// ---------------------------------------------------------------------------
// host: the column plans of a launch. A tile row can take the x half of its coordinates
// from a per-column table when the job is 'ray = B * c0 + A' without normalisation on a
// lat/lon source and the x and z components of A and B are the same for its 8 rows (then
// rx, rz and everything derived from them alone are functions of the column). Tile rows
// with the same four constants share a table.
// ---------------------------------------------------------------------------
extern "C" int eu_current_slot(void);
namespace {
struct plan_cache {
  std::vector<unsigned char> key;
  int *tileplan = nullptr; size_t tileplan_cap = 0;
  float *coltab = nullptr; size_t coltab_cap = 0;
  float *atab = nullptr;
  int planned_rows = 0;      // tile rows with a column plan
  int *l2_rows = nullptr; size_t l2_cap = 0;
  int l2_off[9] = {};
  int *l1_rows = nullptr; size_t l1_cap = 0;
  int l1_off[9] = {};
  hipStream_t last_stream = nullptr;   // where the plans were last read
} g4s[EU_MAX_SLOTS];
// one cache per device slot (eu_api.hip: eu_hip_init_devices)
#define g4 (g4s[eu_current_slot()])

bool ensure_atab()
{
  if (g4.atab) return true;
  std::vector<float> tab(768, 0.0f);
  for (int i = 0; i < EU_ATAN_TAB_ENTRIES; i++) eu_atan_tab_entry(i, tab.data() + 8 * i);
  if (hipMalloc((void **)&g4.atab, tab.size() * sizeof(float)) != hipSuccess) return false;
  return hipMemcpy(g4.atab, tab.data(), tab.size() * sizeof(float), hipMemcpyHostToDevice) == hipSuccess;
}

template <int DEG>
void launch_colplan(const eu_render_params &p, float *ct, const float *k, hipStream_t st)
{
  hipLaunchKernelGGL((eu_colplan_kernel<DEG>), dim3((unsigned)((p.width + 511) / 512)), dim3(256), 0, st, p, ct,
                     k[0], k[1], k[2], k[3]);
}
}  // namespace

// h_row: the host copy of the plan's row table (whole frame), plan_gen: changes whenever the
// stepper tables change. Returns 1 when the job is outside this kernel's coverage (the caller
// goes on to eu_launch_render2 / eu_launch_render).
// only_if_worth: take a lat/lon job only where the staged kernels measured faster than the direct-gather ones
// (cubic / quadratic, the FAST profile, column plans on at least half of the tile rows: an upright cubemap or
// rectilinear target); 0: every job they cover (EU_HIP_R4=1: tests, A/B runs)
extern "C" int eu_launch_render4(const eu_render_params *pp, const float *h_row, size_t h_row_floats,
                                 unsigned long long plan_gen, int only_if_worth, void *stream)
{
  eu_render_params p = *pp;
  if (p.twine || p.stage != 0 || p.form >= EU_FORM_FISH || p.src.has_lcp || p.nch_out != p.nch) return 1;
  if (p.norm_mode != EU_NORM_NONE && p.norm_mode != EU_NORM_DIV) return 1;
  if (p.src.prj != EU_SPHERICAL && p.src.prj != EU_CUBEMAP && p.src.prj != EU_BIATAN6) return 1;
  if (p.src.degree < 1 || p.src.degree > 3 || p.src.es0 != p.nch) return 1;
  if (p.nch != 3 && p.nch != 4) return 1;
  if (!p.wl) return 1;
  // the staging offsets are 32-bit
  if (p.src.es1 * 4 >= (1ll << 31)) return 1;
  p.tiles_y = (p.row_end - p.row_begin + EU4_TH - 1) / EU4_TH;
  eu4_plan w;
  w.tiles16 = (p.width + EU4_TW - 1) / EU4_TW;
  if (w.tiles16 <= 0 || p.tiles_y <= 0) return 0;
  if (p.tiles_y > 65535 * 8 * EU4_UNIT_ROWS) return 1;
  hipStream_t st = (hipStream_t)stream;
  if (!ensure_atab()) return -1;
  w.atab_g = g4.atab;

  // ---- column plans (cached while nothing they depend on changes) ----------------------
  std::vector<unsigned char> key(sizeof(unsigned long long) + sizeof(eu_src_dev) + 8 * sizeof(int));
  {
    unsigned char *q = key.data();
    memcpy(q, &plan_gen, sizeof plan_gen); q += sizeof plan_gen;
    eu_src_dev sd = p.src; sd.base = nullptr;
    memcpy(q, &sd, sizeof sd); q += sizeof sd;
    const int v[8] = { p.width, p.row_begin, p.row_end, p.band_shift, p.band_count, p.band_index, p.form, p.norm_mode };
    memcpy(q, v, sizeof v);
  }
  // (cubemap / biatan6 sources never read the tile plan: no plans, no upload, no synchronisation for them)
  if (p.src.prj == EU_SPHERICAL && key != g4.key) {
    std::vector<int> tp((size_t)p.tiles_y, -1);
    std::vector<float> plans;          // 4 floats per plan: A0, A2, B0, B2
    const bool can = p.src.prj == EU_SPHERICAL && p.form == EU_FORM_BA && p.norm_mode == EU_NORM_NONE && h_row;
    static const bool off = [] { const char *e = getenv("EU_HIP_COLPLAN"); return e && e[0] == '0'; }();
    if (can && !off) {
      for (int ty = 0; ty < p.tiles_y; ty++) {
        float k[4] = { 0, 0, 0, 0 };
        bool same = true;
        for (int ly = 0; ly < EU4_TH && same; ly++) {
          const int y = std::min(p.row_begin + ty * EU4_TH + ly, p.row_end - 1);
          const size_t fr = (size_t)eu_frame_row(y, p.band_shift, p.band_count, p.band_index) * EU_ROW_FLOATS;
          if (fr + 6 > h_row_floats) { same = false; break; }
          const float c[4] = { h_row[fr + 0], h_row[fr + 2], h_row[fr + 3], h_row[fr + 5] };
          if (ly == 0) memcpy(k, c, sizeof k);
          else same = memcmp(k, c, sizeof k) == 0;
        }
        if (!same) continue;
        int id = -1;
        for (size_t j = 0; j < plans.size() / 4 && id < 0; j++)
          if (memcmp(&plans[4 * j], k, sizeof k) == 0) id = (int)j;
        if (id < 0 && plans.size() / 4 < EU4_MAX_PLANS) {
          id = (int)(plans.size() / 4);
          plans.insert(plans.end(), k, k + 4);
        }
        tp[(size_t)ty] = id;
      }
    }
    if (g4.tileplan_cap < tp.size()) {
      if (g4.tileplan) (void)hipFree(g4.tileplan);
      g4.tileplan = nullptr; g4.tileplan_cap = 0;
      if (hipMalloc((void **)&g4.tileplan, tp.size() * sizeof(int)) != hipSuccess) return -1;
      g4.tileplan_cap = tp.size();
    }
    const size_t need = std::max<size_t>(1, plans.size() / 4) * (size_t)p.width * EU4_COL_FLOATS;
    if (g4.coltab_cap < need) {
      if (g4.coltab) (void)hipFree(g4.coltab);
      g4.coltab = nullptr; g4.coltab_cap = 0;
      if (hipMalloc((void **)&g4.coltab, need * sizeof(float)) != hipSuccess) return -1;
      g4.coltab_cap = need;
    }
    // the previous launch may still read the old plans - on this stream or on the one the plans were last
    // used on (the library's own stream and a caller's stream alternate)
    if (hipStreamSynchronize(st) != hipSuccess) return -1;
    if (g4.last_stream && g4.last_stream != st && hipStreamSynchronize(g4.last_stream) != hipSuccess) return -1;
    if (hipMemcpy(g4.tileplan, tp.data(), tp.size() * sizeof(int), hipMemcpyHostToDevice) != hipSuccess) return -1;
    g4.planned_rows = 0;
    for (int v : tp) g4.planned_rows += v >= 0;
    {
      // the second loop's rows per XCD: units of EU5_UNIT_ROWS tile rows dealt round-robin, as in the first loop
      std::vector<int> rows[8];
      for (int r = 0; r < p.tiles_y; r++) {
        const int m = r >> 1;
        const bool paired = 2 * m + 1 < p.tiles_y && tp[(size_t)2 * m] >= 0 && tp[(size_t)2 * m] == tp[(size_t)2 * m + 1];
        if (!paired) rows[(r / EU5_UNIT_ROWS) & 7].push_back(r);
      }
      std::vector<int> all;
      for (int x = 0; x < 8; x++) { g4.l2_off[x] = (int)all.size(); all.insert(all.end(), rows[x].begin(), rows[x].end()); }
      g4.l2_off[8] = (int)all.size();
      if (g4.l2_cap < all.size() + 1) {
        if (g4.l2_rows) (void)hipFree(g4.l2_rows);
        g4.l2_rows = nullptr; g4.l2_cap = 0;
        if (hipMalloc((void **)&g4.l2_rows, (all.size() + 1) * sizeof(int)) != hipSuccess) return -1;
        g4.l2_cap = all.size() + 1;
      }
      if (!all.empty() && hipMemcpy(g4.l2_rows, all.data(), all.size() * sizeof(int), hipMemcpyHostToDevice) != hipSuccess) return -1;
      // the first loop's double rows per XCD (units of EU5_UNIT_ROWS / 2 double rows), with their plans
      std::vector<int> prs[8];
      for (int m = 0; 2 * m + 1 < p.tiles_y; m++)
        if (tp[(size_t)2 * m] >= 0 && tp[(size_t)2 * m] == tp[(size_t)2 * m + 1]) {
          auto &v = prs[(m / (EU5_UNIT_ROWS / 2)) & 7];
          v.push_back(m); v.push_back(tp[(size_t)2 * m]);
        }
      std::vector<int> all1;
      for (int x = 0; x < 8; x++) { g4.l1_off[x] = (int)all1.size() / 2; all1.insert(all1.end(), prs[x].begin(), prs[x].end()); }
      g4.l1_off[8] = (int)all1.size() / 2;
      if (g4.l1_cap < all1.size() + 2) {
        if (g4.l1_rows) (void)hipFree(g4.l1_rows);
        g4.l1_rows = nullptr; g4.l1_cap = 0;
        if (hipMalloc((void **)&g4.l1_rows, (all1.size() + 2) * sizeof(int)) != hipSuccess) return -1;
        g4.l1_cap = all1.size() + 2;
      }
      if (!all1.empty() && hipMemcpy(g4.l1_rows, all1.data(), all1.size() * sizeof(int), hipMemcpyHostToDevice) != hipSuccess) return -1;
    }
    for (size_t j = 0; j < plans.size() / 4; j++) {
      float *ct = g4.coltab + j * (size_t)p.width * EU4_COL_FLOATS;
      switch (p.src.degree) {
        case 1: launch_colplan<1>(p, ct, &plans[4 * j], st); break;
        case 2: launch_colplan<2>(p, ct, &plans[4 * j], st); break;
        default: launch_colplan<3>(p, ct, &plans[4 * j], st); break;
      }
      if (hipGetLastError() != hipSuccess) return -1;
    }
    g4.key.swap(key);
  }
  if (only_if_worth && p.src.prj == EU_SPHERICAL) {
    if (p.src.degree < 2 || g4.planned_rows * 2 < p.tiles_y) return 1;
  }
  w.tileplan = g4.tileplan;
  w.coltab = g4.coltab;
  w.l2_rows = g4.l2_rows;
  for (int x = 0; x < 9; x++) w.l2_off[x] = g4.l2_off[x];
  w.l2_half = (w.tiles16 + 1) / 2;
  w.l2_magic = (1ull << 40) / (unsigned long long)w.l2_half + 1;
  w.l1_rows = g4.l1_rows;
  for (int x = 0; x < 9; x++) w.l1_off[x] = g4.l1_off[x];
  w.l1_magic = (1ull << 40) / (unsigned long long)w.tiles16 + 1;
  g4.last_stream = st;
#ifdef EU5_STAMPS
  // diagnostic build: stamps of every tile, averaged per pass count / plan kind after the launch
  static unsigned long long *d_st = nullptr; static size_t st_cap = 0; static int dumps = 0;
  const size_t ntile_st = (size_t)w.tiles16 * p.tiles_y * 8;
  const size_t nst = ntile_st + 8192 * 4;      // + per-wave totals (eu_render5_kernel)
  if (st_cap < nst) { if (d_st) (void)hipFree(d_st); if (hipMalloc((void **)&d_st, nst * 8) != hipSuccess) return -1; st_cap = nst; }
  (void)hipMemsetAsync(d_st, 0, nst * 8, st);
  w.stamps = d_st;
  int rc_ = p.nch == 3 ? launch4_n<3>(p, w, st) : p.nch == 4 ? launch4_n<4>(p, w, st) : 1;
  if (rc_ == 0 && dumps < 2 && nst >= 8 * 4096) {
    dumps++;
    std::vector<unsigned long long> h(nst);
    (void)hipStreamSynchronize(st);
    (void)hipMemcpy(h.data(), d_st, nst * 8, hipMemcpyDeviceToHost);
    double acc[32][8] = {}; size_t cnt[32] = {};
    for (size_t t = 0; t < ntile_st / 8; t++) {
      const unsigned long long *q = &h[t * 8];
      if (!q[0] || !q[7]) continue;
      const int cls = (int)(q[1] & 31);
      cnt[cls]++;
      acc[cls][0] += (double)(q[2] - q[0]); acc[cls][1] += (double)(q[3] - q[2]);
      if (q[4]) { acc[cls][2] += (double)(q[4] - q[3]); acc[cls][3] += (double)(q[5] - q[4]); acc[cls][4] += (double)(q[6] - q[5]); }
      acc[cls][5] += (double)(q[7] - q[6]); acc[cls][6] += (double)(q[7] - q[0]);
    }
    {
      double mn = 1e30, mx = 0, sum = 0, l1 = 0, first = 1e30, last = 0; size_t nw = 0;
      for (size_t k = 0; k < 8192; k++) {
        const unsigned long long *q = &h[ntile_st + k * 4];
        if (!q[3]) continue;
        const double d = (double)(q[2] - q[0]);
        mn = std::min(mn, d); mx = std::max(mx, d); sum += d; l1 += (double)(q[1] - q[0]); nw++;
        first = std::min(first, (double)q[0]); last = std::max(last, (double)q[2]);
      }
      if (nw) fprintf(stderr, "eu5 waves: %zu, cycles per wave min %.0f avg %.0f max %.0f, first loop avg %.0f, first start .. last end %.0f\n",
                      nw, mn, sum / nw, mx, l1 / nw, last - first);
    }
    for (int c = 0; c < 32; c++) if (cnt[c])
      fprintf(stderr, "eu5 stamps: hoist %d npass %2d tiles %8zu | coords %7.0f box %6.0f dma-issue+weights %6.0f dma-wait %6.0f taps(all passes) %6.0f store %5.0f | tile %7.0f (100 MHz ticks? s_memtime)\n",
              c >> 4, (c & 15) - 1, cnt[c], acc[c][0] / cnt[c], acc[c][1] / cnt[c], acc[c][2] / cnt[c], acc[c][3] / cnt[c], acc[c][4] / cnt[c], acc[c][5] / cnt[c], acc[c][6] / cnt[c]);
  }
  return rc_;
#else
  switch (p.nch) {
    case 3: return launch4_n<3>(p, w, st);
    case 4: return launch4_n<4>(p, w, st);
  }
  return 1;
#endif
}

// included by eu_render4.hip behind eu4_plan, eu4_dma_row and the work-list layout (EU4_WL_*)
// ---------------------------------------------------------------------------
// Round 3: the staged kernel as PERSISTENT wavefronts (eu_render5_kernel).
//
// What round 2's measurements said about eu_render4s_kernel: with the L1 traffic
// gone a wave lives ~11k cycles of which it issues vector instructions for ~2.5k, and the 16
// single-wave workgroups a CU admits leave four such waves per SIMD; 17 % of the headline's tiles
// (the polar faces' wide boxes) fall to the direct-gather kernel, which takes 0.63 ms for them.
// This kernel keeps the staging and changes what surrounds it:
//   * workgroups of four independent waves stay resident for the whole launch and walk the
//     tiles of their XCD in raster order (wave k of the XCD takes tiles k, k + K, ...): no
//     wave launch, no atanf table load and no LDS allocation per tile; five waves per SIMD
//     (96 registers, 7 KB of LDS per wave + one table per workgroup),
//   * lane -> pixel mapping [half | row | pair]: lanes 0-31 are the left 8x8 half of the 16x8
//     tile, DPP rows of 16 lanes are 8x4 quarters, so ONE reduction yields the boxes of the
//     quarters, the halves and the tile. A tile whose box exceeds the slice is staged and
//     evaluated as two halves or four quarters, one after the other in the same slice
//     (the polar faces of the headline: 60 % of the tiles in one pass, 28 % in two, 8 % in
//     four; 4.7 % - the pole itself - remain for the work list),
//   * the weighted sum written on register PAIRS: ds_read_b128 returns (R,G),(B,X); the
//     x weights of the lane's two pixels sit in one pair and are broadcast with op_sel, so
//     a window row is 14 packed operations per pixel (zimt/eval.h:904-1059: same products,
//     same order of additions, per channel).
// ---------------------------------------------------------------------------
#define EU5_WAVES 4
#ifndef EU5_TEXELS
#define EU5_TEXELS 576      // LDS texels (16 bytes) per wave: 4 x 9 KB + 3 KB table = 39 KB per workgroup, 4 per CU
#endif
#ifndef EU5_OCC
#define EU5_OCC 4           // waves per SIMD the registers are capped for (128: at 96 the tile code spills, and every scratch access costs more than the fifth wave gives)
#endif
#ifndef EU5_UNIT_ROWS
#define EU5_UNIT_ROWS 4     // tile rows per XCD unit
#endif

// the work list a tile goes to: a multiplicative hash of the tile id. (id % EU4_SHARDS keeps the
// tile COLUMN: the tiles around a pole then land in a sixth of the lists, and the direct-gather
// kernel's waves on those lists work through ~10 tiles each while the others idle: 0.23 ms for
// 1.6 % of the headline's tiles.)
__device__ __forceinline__ int eu4_shard_of(int id)
{
  return (int)(((unsigned)id * 0x9E3779B1u) >> 22) & (EU4_SHARDS - 1);
}

typedef const __attribute__((address_space(3))) eu4_f4 *eu5_l4ptr;

// the (d+1)^2 taps of both pixels from the LDS image; a, b: float offsets of the two windows
// inside the wave's slice, pitch in floats. wx[i] / wy[j] = (weight of pixel a, weight of pixel b).
// Results: (R,G) and (B,X) of pixel a and of pixel b.
template <int NCH, int DEG>
__device__ __forceinline__ void eu5_taps(eu_lptr lt, int a, int b, int pitch, const eu_f2 *wx, const eu_f2 *wy,
                                         eu_f2 tx, eu_f2 ty, eu_f2 &rga, eu_f2 &bxa, eu_f2 &rgb, eu_f2 &bxb)
{
  constexpr int order = DEG + 1;
  if constexpr (DEG == 1) {
    // _eval_linear, eval.h:1014-1059: wl = 1 - t, wr = t
    const eu_f2 wl0 = 1.0f - tx, wr0 = tx, wl1 = 1.0f - ty, wr1 = ty;
    const eu4_f4 a00 = *(eu5_l4ptr)(lt + a), a01 = *(eu5_l4ptr)(lt + a + 4);
    const eu4_f4 a10 = *(eu5_l4ptr)(lt + a + pitch), a11 = *(eu5_l4ptr)(lt + a + pitch + 4);
    const eu4_f4 b00 = *(eu5_l4ptr)(lt + b), b01 = *(eu5_l4ptr)(lt + b + 4);
    const eu4_f4 b10 = *(eu5_l4ptr)(lt + b + pitch), b11 = *(eu5_l4ptr)(lt + b + pitch + 4);
    {
      const eu_f2 l0 = { wl0.x, wl0.x }, r0 = { wr0.x, wr0.x }, l1 = { wl1.x, wl1.x }, r1 = { wr1.x, wr1.x };
      eu_f2 s = a00.xy * l0; s = s + a01.xy * r0; s = s * l1;
      eu_f2 u = a10.xy * l0; u = u + a11.xy * r0; rga = s + u * r1;
      s = a00.zw * l0; s = s + a01.zw * r0; s = s * l1;
      u = a10.zw * l0; u = u + a11.zw * r0; bxa = s + u * r1;
    }
    {
      const eu_f2 l0 = { wl0.y, wl0.y }, r0 = { wr0.y, wr0.y }, l1 = { wl1.y, wl1.y }, r1 = { wr1.y, wr1.y };
      eu_f2 s = b00.xy * l0; s = s + b01.xy * r0; s = s * l1;
      eu_f2 u = b10.xy * l0; u = u + b11.xy * r0; rgb = s + u * r1;
      s = b00.zw * l0; s = s + b01.zw * r0; s = s * l1;
      u = b10.zw * l0; u = u + b11.zw * r0; bxb = s + u * r1;
    }
  } else {
#pragma unroll
    for (int j = 0; j < order; j++) {
      eu4_f4 ta[order], tb[order];
#pragma unroll
      for (int i = 0; i < order; i++) {
        ta[i] = *(eu5_l4ptr)(lt + a + j * pitch + 4 * i);
        tb[i] = *(eu5_l4ptr)(lt + b + j * pitch + 4 * i);
      }
      const eu_f2 w0a = { wx[0].x, wx[0].x }, w0b = { wx[0].y, wx[0].y };
      eu_f2 ra = ta[0].xy * w0a, qa = ta[0].zw * w0a, rb = tb[0].xy * w0b, qb = tb[0].zw * w0b;
#pragma unroll
      for (int i = 1; i < order; i++) {
        const eu_f2 wa = { wx[i].x, wx[i].x }, wb = { wx[i].y, wx[i].y };
        ra = ra + wa * ta[i].xy; qa = qa + wa * ta[i].zw;
        rb = rb + wb * tb[i].xy; qb = qb + wb * tb[i].zw;
      }
      const eu_f2 ya = { wy[j].x, wy[j].x }, yb = { wy[j].y, wy[j].y };
      if (j == 0) { rga = ra * ya; bxa = qa * ya; rgb = rb * yb; bxb = qb * yb; }
      else { rga = rga + ra * ya; bxa = bxa + qa * ya; rgb = rgb + rb * yb; bxb = bxb + qb * yb; }
    }
  }
}

// Bounding boxes by one DPP reduction: log-step row shifts inside the rows of 16 lanes (lane 15
// of every row then holds its quarter's box), then the row broadcast into COPIES (lanes 31 and
// 63 of the copies hold the boxes of the halves; the quarters stay readable). Four reductions
// interleaved: three independent instructions between dependent DPP operations.
__device__ __forceinline__ void eu5_box_reduce(int &q0, int &q1, int &q2, int &q3, int &h0, int &h1, int &h2, int &h3)
{
#define EU5_RED(ctrl)                                        \
  "v_min_i32_dpp %0, %0, %0 " ctrl "\n\t"                    \
  "v_min_i32_dpp %1, %1, %1 " ctrl "\n\t"                    \
  "v_max_i32_dpp %2, %2, %2 " ctrl "\n\t"                    \
  "v_max_i32_dpp %3, %3, %3 " ctrl "\n\t"
  asm("s_nop 1\n\t"
      EU5_RED("row_shr:1 row_mask:0xf bank_mask:0xf")
      EU5_RED("row_shr:2 row_mask:0xf bank_mask:0xf")
      EU5_RED("row_shr:4 row_mask:0xf bank_mask:0xf")
      EU5_RED("row_shr:8 row_mask:0xf bank_mask:0xf")
      "v_mov_b32 %4, %0\n\tv_mov_b32 %5, %1\n\tv_mov_b32 %6, %2\n\tv_mov_b32 %7, %3\n\t"
      "v_min_i32_dpp %4, %0, %4 row_bcast:15 row_mask:0xa bank_mask:0xf\n\t"
      "v_min_i32_dpp %5, %1, %5 row_bcast:15 row_mask:0xa bank_mask:0xf\n\t"
      "v_max_i32_dpp %6, %2, %6 row_bcast:15 row_mask:0xa bank_mask:0xf\n\t"
      "v_max_i32_dpp %7, %3, %7 row_bcast:15 row_mask:0xa bank_mask:0xf\n\t"
      "s_nop 0"
      : "+v"(q0), "+v"(q1), "+v"(q2), "+v"(q3), "=&v"(h0), "=&v"(h1), "=&v"(h2), "=&v"(h3));
#undef EU5_RED
}

struct eu5_box { int mnx, mny, mxx, mxy; };

// for a box width w (1..64): ceil(2^16 / w) in the low 17 bits - (lane * M) >> 16 is lane / w for every
// lane < 64 - and 64 / w, the rows one LDS-DMA instruction covers, in the top byte
struct eu5_divtab_t { unsigned v[65]; };
static constexpr eu5_divtab_t eu5_make_divtab()
{
  eu5_divtab_t t = {};
  t.v[0] = 0;
  for (unsigned w = 1; w <= 64; w++) t.v[w] = ((65536u + w - 1) / w) | ((64u / w) << 24);
  return t;
}
__constant__ const eu5_divtab_t eu5_divtab = eu5_make_divtab();

__device__ __forceinline__ eu5_box eu5_box_at(int a, int b, int c, int d, int lane)
{
  eu5_box r;
  r.mnx = __builtin_amdgcn_readlane(a, lane); r.mny = __builtin_amdgcn_readlane(b, lane);
  r.mxx = __builtin_amdgcn_readlane(c, lane); r.mxy = __builtin_amdgcn_readlane(d, lane);
  return r;
}
__device__ __forceinline__ eu5_box eu5_box_join(const eu5_box &a, const eu5_box &b)
{
  eu5_box r = { min(a.mnx, b.mnx), min(a.mny, b.mny), max(a.mxx, b.mxx), max(a.mxy, b.mxy) };
  return r;
}
// 1: fits the slice, 0: does not, -1: empty (no hitting pixel)
template <int ORDER>
__device__ __forceinline__ int eu5_box_fits(const eu5_box &b)
{
  if (b.mnx == INT_MAX) return -1;
  // 32-bit scalar arithmetic: base positions are gated into the core (lanes whose coordinate the gate
  // would have had to fold make the tile unclean, whatever this says)
  const unsigned bw = (unsigned)b.mxx - (unsigned)b.mnx + ORDER, bh = (unsigned)b.mxy - (unsigned)b.mny + ORDER;
  return bw <= 64u && bh <= (unsigned)EU5_TEXELS && bw * bh <= (unsigned)EU5_TEXELS;
}

// gate arithmetic without the per-lane range test (map.h:341-357, :423-440): a coordinate that needs
// folding ends up outside the core, which eu5_tile reads off the tile's box with scalar compares
__device__ __forceinline__ eu_f2 eu5_gate2(eu_f2 c, int kind, float lower, float upper)
{
  if (kind == 0) {            // clamp_gate, map.h:231-236
    eu_f2 r = c;
    r = eu_sel2(c < lower, (eu_f2){ lower, lower }, r);
    r = eu_sel2(c > upper, (eu_f2){ upper, upper }, r);
    return r;
  }
  eu_f2 cc = c - lower;
  if (kind == 1) cc = eu_abs2(cc);
  return cc + lower;
}

// -1 for a lane whose |a|, |b|, |c| are not all inside [2^-40, 2^40] (the range the FMA division
// and square root sequences are used in; a NaN cannot occur: the stepper tables are finite)
__device__ __forceinline__ int eu5_out_of_range3(float a, float b, float c)
{
  const float hi = __builtin_fmaxf(__builtin_fmaxf(__builtin_fabsf(a), __builtin_fabsf(b)), __builtin_fabsf(c));
  const float lo = __builtin_fminf(__builtin_fminf(__builtin_fabsf(a), __builtin_fabsf(b)), __builtin_fabsf(c));
  return (hi <= 0x1p40f && lo >= 0x1p-40f) ? 0 : -1;
}

// FAST: the job's uniform switches as compile-time constants - 'ray = B * c0 + A' without
// normalisation, no bands, every ray hits, brighten 1 (what a cubemap / rectilinear target of a
// full-sphere or cubemap source is). The persistent loop keeps every scalar it uses live: fewer of
// them means fewer SGPR spills in the tile code.
//
// The tile code is cut into stages so that the loop can run them SKEWED (software pipelining over
// the tiles of a wave):   load(t+1) | box(t) | DMA issue(t) | coordinates(t+1) | taps(t), store(t)
// The LDS-DMA of tile t is in flight under the coordinate chain of tile t+1 (~2-7k cycles of vector
// work: the round trip to HBM hides completely), the table loads of tile t+1 under the box
// reduction of tile t. vmcnt counts in issue order, so nothing that tile t+1's chain waits for may be
// younger than tile t's DMA: its loads are issued before it and consumed (eu5_settle) before it.

struct eu5_loaded {           // what the table loads of a tile return (20 registers, two readings)
  eu4_f4 a0, a1, b0, b1;      // HOIST: the column entries of the lane's two pixels
                              // otherwise: a0 = (A0, A2, B0, B2), a1 = (C0, C1, C2, -), b0.xy = c1
  eu_f2 c0;                   // column table of the stepper
  float A1, B1;               // row constants of the lane's row
  int valid;                  // bit 3 / 4: pixel a / b lies inside the frame
};

template <int DEG, bool HOIST>
struct eu5_coord {            // what the coordinate stage hands to the evaluation stage
  int ixa, ixb, iya, iyb;     // base positions (split: basis.h:102-146)
  eu_f2 ty;                   // fractional part
  eu_f2 wx[(HOIST && DEG >= 2) ? DEG + 1 : 1];   // HOIST, degree >= 2: the x weights (pixel a, pixel b) of the column
                              // table; otherwise wx[0] = tx (the weights are formed by the evaluation stage)
  int flags;                  // bit 0 / 1: pixel a / b hits, bit 2: a hitting pixel left the fast path,
                              // bit 3 / 4: pixel a / b lies inside the frame
};

// the consumer the compiler has to see before the DMA is issued
__device__ __forceinline__ void eu5_settle(eu5_loaded &L)
{
  asm volatile("" : "+v"(L.a0), "+v"(L.a1), "+v"(L.b0), "+v"(L.b1), "+v"(L.c0), "+v"(L.A1), "+v"(L.B1));
}

// lane -> pixels of the tile: [half | row | pair]
struct eu5_where { int xa, xb, xac, xbc, yc; bool va, vb; };
__device__ __forceinline__ eu5_where eu5_locate(const eu_render_params &p, int tile_y, int x0, int lane)
{
  eu5_where q;
  const int pr = lane & 3, rw = (lane >> 2) & 7, hf = lane >> 5;
  const int y = p.row_begin + tile_y * EU4_TH + rw;
  const bool yin = y < p.row_end;
  q.yc = yin ? y : p.row_end - 1;
  q.xa = x0 + 8 * hf + 2 * pr; q.xb = q.xa + 1;
  q.va = yin && q.xa < p.width; q.vb = yin && q.xb < p.width;
  q.xac = q.xa < p.width ? q.xa : p.width - 1; q.xbc = q.xb < p.width ? q.xb : p.width - 1;
  return q;
}

template <bool HOIST, bool FAST>
__device__ __forceinline__ void eu5_load(const eu_render_params &p, const float *ct, int tile_y, int x0, int lane, eu5_loaded &L)
{
  const eu5_where q = eu5_locate(p, tile_y, x0, lane);
  const float *rt = p.row + (long long)(FAST ? q.yc : eu_frame_row(q.yc, p.band_shift, p.band_count, p.band_index)) * EU_ROW_FLOATS;
  L.c0 = (eu_f2){ p.col[q.xac], p.col[q.xbc] };
  L.A1 = rt[1]; L.B1 = rt[4];
  L.valid = (q.va ? 8 : 0) | (q.vb ? 16 : 0);
  if constexpr (HOIST) {
    const eu4_f4 *ea = (const eu4_f4 *)(ct + (size_t)q.xac * EU4_COL_FLOATS);
    const eu4_f4 *eb = (const eu4_f4 *)(ct + (size_t)q.xbc * EU4_COL_FLOATS);
    L.a0 = ea[0]; L.a1 = ea[1]; L.b0 = eb[0]; L.b1 = eb[1];
  } else {
    L.a0 = (eu4_f4){ rt[0], rt[2], rt[3], rt[5] };
    if (!FAST && p.form == EU_FORM_BCA) {
      L.a1 = (eu4_f4){ rt[6], rt[7], rt[8], 0.0f };
      const float *colB = p.col + p.width;
      L.b0.x = colB[q.xac]; L.b0.y = colB[q.xbc];
    }
  }
}

template <int NCH, int DEG, int PRJ, bool HOIST, bool FAST>
__device__ __forceinline__ void eu5_coords(const eu_render_params &p, const float *atab, const eu5_loaded &L,
                                           int tile_y, int x0, int lane, eu5_coord<DEG, HOIST> &C)
{
  const eu_src_dev &s = p.src;
  eu_f2 gy, tx;
  eu_i2 hit = { -1, -1 }, ok = { -1, -1 };
  constexpr bool LEAN = FAST && PRJ == EU_SPHERICAL;
  if constexpr (HOIST) {
    C.ixa = __float_as_int(L.a0.x); C.ixb = __float_as_int(L.b0.x);
    tx = (eu_f2){ L.a0.y, L.b0.y };
    if constexpr (DEG >= 2) {
      C.wx[0] = (eu_f2){ L.a0.z, L.b0.z }; C.wx[1] = (eu_f2){ L.a0.w, L.b0.w }; C.wx[2] = (eu_f2){ L.a1.x, L.b1.x };
      if constexpr (DEG == 3) C.wx[3] = (eu_f2){ L.a1.y, L.b1.y };
    }
  }
  if constexpr (LEAN) {
    // the reference's operations in the reference's order (stepper.h ray, geometry.h:278-301,
    // environment.h:988-1006, map.h gates) in their leanest instruction forms (eu_math2.h, round 3)
    eu_i2 big0 = { 0, 0 }, big1 = { 0, 0 };
    eu_f2 lat;
    if constexpr (HOIST) {
      const eu_f2 ryy = L.B1 * L.c0 + L.A1;
      const eu_f2 qs = { L.a1.z, L.b1.z };
      ok = (eu_i2){ (C.ixa != INT_MIN ? -1 : 0) & ~eu5_out_of_range3(ryy.x, qs.x, qs.x),
                    (C.ixb != INT_MIN ? -1 : 0) & ~eu5_out_of_range3(ryy.y, qs.y, qs.y) };
      lat = eu_atan2f_2_lean(ryy, qs, atab, 1, big0);
    } else {
      const float A0 = L.a0.x, A2 = L.a0.y, B0 = L.a0.z, B2 = L.a0.w;
      const eu_f2 rx = B0 * L.c0 + A0, ry = L.B1 * L.c0 + L.A1, rz = B2 * L.c0 + A2;
      ok = (eu_i2){ ~eu5_out_of_range3(rx.x, ry.x, rz.x), ~eu5_out_of_range3(rx.y, ry.y, rz.y) };
      const eu_f2 q2 = rx * rx + rz * rz;
      const eu_f2 qs = eu_sqrt2_safe(q2);
      lat = eu_atan2f_2_lean(ry, qs, atab, 1, big0);
      // one chain after the other: interleaved they need more registers than the loop has to spare
      __builtin_amdgcn_sched_barrier(0);
      const eu_f2 lon = eu_atan2f_2_lean(rx, rz, atab, 0, big1);
      eu_f2 i0 = { (float)((double)lon.x - s.tex_x0), (float)((double)lon.y - s.tex_x0) };
      if (s.cdiv_ok) i0 = eu_div2_const(i0, s.ext_w, s.rcp_ext_w);
      else i0 = eu_div2_rr(i0, s.ext_w, eu_rcp_refined(s.ext_w));
      i0 = i0 * s.total_w; i0 = i0 - .5f;
      const eu_f2 sx = i0 - s.win_x_off;
      const eu_f2 gx = eu5_gate2(sx, s.gate0, s.lower0, s.upper0);
      eu_f2 fx;
      if constexpr (DEG & 1) fx = (eu_f2){ floorf(gx.x), floorf(gx.y) };
      else fx = (eu_f2){ roundf(gx.x), roundf(gx.y) };
      tx = gx - fx;
      C.ixa = (int)fx.x; C.ixb = (int)fx.y;
    }
    ok = ok & ~(big0 | big1);
    eu_f2 i1 = { (float)((double)lat.x - s.tex_y0), (float)((double)lat.y - s.tex_y0) };
    if (s.cdiv_ok) i1 = eu_div2_const(i1, s.ext_h, s.rcp_ext_h);
    else i1 = eu_div2_rr(i1, s.ext_h, eu_rcp_refined(s.ext_h));
    i1 = i1 * s.total_h; i1 = i1 - .5f;
    const eu_f2 sy = i1 - s.win_y_off;
    gy = eu5_gate2(sy, s.gate1, s.lower1, s.upper1);
  } else if constexpr (HOIST) {
    const eu_f2 ryy = L.B1 * L.c0 + L.A1;
    ok = ok & (eu_i2){ C.ixa != INT_MIN ? -1 : 0, C.ixb != INT_MIN ? -1 : 0 };
    const eu_f2 qs = { L.a1.z, L.b1.z };
    const eu_f2 lat = eu_atan2f_2_tab_ok(ryy, qs, atab, 1, ok);
    if (!FAST && !s.always_hit) {
      const eu_f2 lon = { L.a1.w, L.b1.w };
      hit = (lon >= s.wex0) & (lon <= s.wex1) & (lat >= s.wex2) & (lat <= s.wex3);
    }
    eu_f2 i1 = { (float)((double)lat.x - s.tex_y0), (float)((double)lat.y - s.tex_y0) };
    if (s.cdiv_ok) i1 = eu_div2_const(i1, s.ext_h, s.rcp_ext_h);
    else i1 = i1 / s.ext_h;
    i1 = i1 * s.total_h; i1 = i1 - .5f;
    const eu_f2 sy = i1 - s.win_y_off;
    gy = eu_gate2_ok(sy, s.gate1, s.lower1, s.upper1, ok);
  } else {
    eu_ray2 r;
    const float A0 = L.a0.x, A2 = L.a0.y, B0 = L.a0.z, B2 = L.a0.w;
    if (!FAST && p.form == EU_FORM_BCA) {
      const float C0 = L.a1.x, C1 = L.a1.y, C2 = L.a1.z;
      const eu_f2 c1 = { L.b0.x, L.b0.y };
      r.x = B0 * L.c0 + C0 * c1 + A0;
      r.y = L.B1 * L.c0 + C1 * c1 + L.A1;
      r.z = B2 * L.c0 + C2 * c1 + A2;
    } else {
      r.x = B0 * L.c0 + A0;
      r.y = L.B1 * L.c0 + L.A1;
      r.z = B2 * L.c0 + A2;
    }
    if (!FAST && p.norm_mode == EU_NORM_DIV) {
      eu_f2 sqn = r.x * r.x; sqn = sqn + r.y * r.y; sqn = sqn + r.z * r.z;
      const eu_f2 n = { sqrtf(sqn.x), sqrtf(sqn.y) };
      r.x = r.x / n; r.y = r.y / n; r.z = r.z / n;
    }
    eu_f2 sx, sy;
    hit = eu_coord2_ok<PRJ, FAST>(s, r, sx, sy, atab, ok);
    const eu_f2 gx = eu_gate2_ok(sx, s.gate0, s.lower0, s.upper0, ok);
    gy = eu_gate2_ok(sy, s.gate1, s.lower1, s.upper1, ok);
    eu_f2 fx;
    if constexpr (DEG & 1) fx = (eu_f2){ floorf(gx.x), floorf(gx.y) };
    else fx = (eu_f2){ roundf(gx.x), roundf(gx.y) };
    tx = gx - fx;
    C.ixa = (int)fx.x; C.ixb = (int)fx.y;
  }
  if constexpr (!HOIST || DEG < 2) C.wx[0] = tx;
  hit = hit & (eu_i2){ (L.valid & 8) ? -1 : 0, (L.valid & 16) ? -1 : 0 };
  eu_f2 fy;
  if constexpr (DEG & 1) fy = (eu_f2){ floorf(gy.x), floorf(gy.y) };
  else fy = (eu_f2){ roundf(gy.x), roundf(gy.y) };
  C.ty = gy - fy;
  C.iya = (int)fy.x; C.iyb = (int)fy.y;
  C.flags = (hit.x ? 1 : 0) | (hit.y ? 2 : 0) | (((hit.x && !ok.x) || (hit.y && !ok.y)) ? 4 : 0) | L.valid;
}

// the boxes of a tile and what follows from them (scalars)
struct eu5_plan1 {
  eu5_box box0;                         // the box of pass 0
  int npass;                            // 0: nothing hits, 1 / 2 / 4: passes, -1: left to the work list
};

// per-lane reduction registers: quarters (q*, lane 15 of every row of 16 lanes), halves (h*, lanes 31, 63)
struct eu5_red { int q0, q1, q2, q3, h0, h1, h2, h3; };

template <int DEG, bool HOIST>
__device__ __forceinline__ void eu5_reduce(const eu5_coord<DEG, HOIST> &C, eu5_red &R)
{
  int q0 = INT_MAX, q1 = INT_MAX, q2 = INT_MIN, q3 = INT_MIN;
  if (C.flags & 1) { q0 = C.ixa; q2 = C.ixa; q1 = C.iya; q3 = C.iya; }
  if (C.flags & 2) { q0 = min(q0, C.ixb); q2 = max(q2, C.ixb); q1 = min(q1, C.iyb); q3 = max(q3, C.iyb); }
  eu5_box_reduce(q0, q1, q2, q3, R.h0, R.h1, R.h2, R.h3);
  R.q0 = q0; R.q1 = q1; R.q2 = q2; R.q3 = q3;
}

__device__ __forceinline__ eu5_box eu5_uniform(eu5_box b)
{
  // wave-uniform by construction; said explicitly, so that what derives from it (LDS-DMA bases, the
  // division table's index) stays scalar
  b.mnx = __builtin_amdgcn_readfirstlane(b.mnx); b.mny = __builtin_amdgcn_readfirstlane(b.mny);
  b.mxx = __builtin_amdgcn_readfirstlane(b.mxx); b.mxy = __builtin_amdgcn_readfirstlane(b.mxy);
  return b;
}

template <int DEG, bool HOIST, bool LEAN>
__device__ __forceinline__ void eu5_boxes(const eu_render_params &p, const eu4_plan &w, const eu5_coord<DEG, HOIST> &C,
                                          int tile_y, int x0, int lane, eu5_plan1 &B)
{
  constexpr int order = DEG + 1;
  const eu_src_dev &s = p.src;
  eu5_red R;
  eu5_reduce<DEG, HOIST>(C, R);
  const eu5_box half0 = eu5_box_at(R.h0, R.h1, R.h2, R.h3, 31), half1 = eu5_box_at(R.h0, R.h1, R.h2, R.h3, 63);
  const eu5_box full = eu5_box_join(half0, half1);
  bool clean = __ballot((C.flags & 4) != 0) == 0ull;
  if constexpr (LEAN) {
    // the gates' range tests, on the box: a coordinate the periodic gate folds (c < lower or
    // c - lower >= width) leaves ix <= -1 or ix >= width - 1, one the mirror gate folds from
    // above ix >= width - 1 (a superset: the tiles on the seam, which do not fit anyway)
    const int cw = (int)(s.upper0 + 0.5f), ch = (int)(s.upper1 + 0.5f);
    if (full.mnx != INT_MAX) {
      if (s.gate0 == 2 && full.mnx < 0) clean = false;
      if (s.gate0 != 0 && full.mxx >= cw - 1) clean = false;
      if (s.gate1 == 2 && full.mny < 0) clean = false;
      if (s.gate1 != 0 && full.mxy >= ch - 1) clean = false;
    }
  }
  // passes: the tile at once, its halves or its quarters, whichever fits the slice first. The boxes of
  // later passes are reduced again when their turn comes: two or four passes are the exception, and
  // eight registers less live across the next tile's coordinate chain are worth more.
  int npass = 1;
  eu5_box b0 = full;
  const int f = eu5_box_fits<order>(full);
  if (f < 0) npass = 0;
  else if (f == 0) {
    npass = 2; b0 = half0;
    if (eu5_box_fits<order>(half0) == 0 || eu5_box_fits<order>(half1) == 0) {
      npass = 4; b0 = eu5_box_at(R.q0, R.q1, R.q2, R.q3, 15);
      if (eu5_box_fits<order>(b0) == 0 || eu5_box_fits<order>(eu5_box_at(R.q0, R.q1, R.q2, R.q3, 31)) == 0 ||
          eu5_box_fits<order>(eu5_box_at(R.q0, R.q1, R.q2, R.q3, 47)) == 0 || eu5_box_fits<order>(eu5_box_at(R.q0, R.q1, R.q2, R.q3, 63)) == 0)
        npass = -1;
    }
  }
  if (npass > 0 && !clean) npass = -1;
  if (npass < 0) {
    // not even the quarters fit (the pole of a lat/lon source, the +-180 degree seam, strong
    // minification), or a hitting pixel left the fast path of the coordinate arithmetic: left to the
    // direct-gather kernel behind this one
    if (lane == 0) {
      const int id = tile_y * w.tiles16 + x0 / EU4_TW;
      const int sh = eu4_shard_of(id);
      const int slot = atomicAdd(p.wl + EU4_WL_SHARD(sh), 1);
      p.wl[EU4_WL_ENTRIES + (size_t)slot * EU4_SHARDS + sh] = id;
    }
  }
  B.npass = __builtin_amdgcn_readfirstlane(npass);
  B.box0 = eu5_uniform(b0);
}

// stage a box: lane L fetches the texel of box column L % ibw in row L / ibw of the k = 64 / ibw rows
// ONE LDS-DMA instruction covers (an LDS-DMA instruction costs its wave 60-180 cycles of issue whatever
// it moves: one per box row was 14-25 per tile); the LDS image is the box, rows back to back
template <int NCH, int DEG>
__device__ __forceinline__ void eu5_stage(const eu_src_dev &s, const eu5_box &bx, unsigned lds_tile, int lane)
{
  constexpr int order = DEG + 1;
  if (bx.mnx == INT_MAX) return;
  const int ibw = bx.mxx - bx.mnx + order, ibh = bx.mxy - bx.mny + order;
  const unsigned tv = eu5_divtab.v[ibw];
  const int k = (int)(tv >> 24);
  const unsigned r = ((unsigned)lane * (tv & 0x1ffffu)) >> 16, c = (unsigned)lane - r * (unsigned)ibw;
  const int bx0 = bx.mnx - DEG / 2, by0 = bx.mny - DEG / 2;
  const unsigned pitchb = (unsigned)(s.es1 * 4);
  const unsigned voff = r * pitchb + c * (NCH * 4u);
  const char *const sb0 = (const char *)(s.base + ((long long)by0 * s.es1 + (long long)bx0 * NCH));
  const unsigned long long step = (unsigned long long)k * pitchb;
  const unsigned dstep = (unsigned)(k * ibw) * 16u;
  if ((int)r < k) {
    const char *sb = sb0;
    unsigned dst = lds_tile;
#pragma unroll 1
    for (int left = ibh; left >= k; left -= k) { eu4_dma_row(dst, voff, sb); sb += step; dst += dstep; }
  }
  if ((int)r < ibh % k) {
    const int full = ibh / k;
    eu4_dma_row(lds_tile + (unsigned)full * dstep, voff, sb0 + (unsigned long long)full * step);
  }
}

// weights of the y axis, the taps of every pass, brighten, store
template <int NCH, int DEG, bool HOIST, bool FAST>
__device__ __forceinline__ void eu5_finish(const eu_render_params &p, const eu5_coord<DEG, HOIST> &C, const eu5_plan1 &B,
                                           float *wtile, int tile_y, int x0, int lane)
{
  constexpr int order = DEG + 1;
  const eu_src_dev &s = p.src;
  const bool hita = C.flags & 1, hitb = C.flags & 2;
  eu_f2 rga = { 0.0f, 0.0f }, bxa = { 0.0f, 0.0f }, rgb = { 0.0f, 0.0f }, bxb = { 0.0f, 0.0f };
  if (B.npass > 0) {
    eu_f2 wy[order], wx[order];
    if constexpr (DEG >= 2) {
      eu_weights2<DEG>(s.wm, C.ty, wy);
      if constexpr (HOIST) {
#pragma unroll
        for (int i = 0; i < order; i++) wx[i] = C.wx[i];
      } else eu_weights2<DEG>(s.wm, C.wx[0], wx);
    }
    const unsigned lds_tile = (unsigned)(unsigned long long)(eu4_lds_void)wtile;
    const int grp = B.npass == 1 ? 0 : B.npass == 2 ? (lane >> 5) : (lane >> 4);
    eu5_box bx = B.box0;                              // pass 0: staged before the next tile's coordinates
#pragma unroll 1
    for (int pi = 0; pi < B.npass; pi++) {
      if (pi > 0) {
        // lane 63 of the halves' copies, lane 31 / 47 / 63 of the quarters
        eu5_red R;
        eu5_reduce<DEG, HOIST>(C, R);
        const int ln = B.npass == 2 ? 63 : 15 + 16 * pi;
        const eu5_box bh = eu5_box_at(R.h0, R.h1, R.h2, R.h3, ln), bq = eu5_box_at(R.q0, R.q1, R.q2, R.q3, ln);
        bx = eu5_uniform(B.npass == 2 ? bh : bq);
        eu5_stage<NCH, DEG>(s, bx, lds_tile, lane);
      }
      if (bx.mnx == INT_MAX) continue;               // a half / quarter without a hitting pixel
      const int ibw = bx.mxx - bx.mnx + order;
      // lanes of other groups and lanes without a hit read the box origin
      const bool mine = grp == pi;
      const int oa = (mine && hita) ? ((C.iya - bx.mny) * ibw + (C.ixa - bx.mnx)) * 4 : 0;
      const int ob = (mine && hitb) ? ((C.iyb - bx.mny) * ibw + (C.ixb - bx.mnx)) * 4 : 0;
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      if (mine) eu5_taps<NCH, DEG>((eu_lptr)wtile, oa, ob, ibw * 4, wx, wy, C.wx[0], C.ty, rga, bxa, rgb, bxb);
    }
  }
  // environment::eval brighten (environment.h:1821-1842), zero on a miss; storer
  float qa[4] = { rga.x, rga.y, bxa.x, bxa.y }, qb[4] = { rgb.x, rgb.y, bxb.x, bxb.y };
  constexpr int ncol = (NCH == 2 || NCH == 4) ? NCH - 1 : NCH;
  const bool bright = !FAST && s.brighten != 1.0f;
#pragma unroll
  for (int c = 0; c < NCH; c++) {
    float a = qa[c], bb = qb[c];
    if (bright && c < ncol) { a = a * s.brighten; bb = bb * s.brighten; }
    qa[c] = hita ? a : 0.0f;
    qb[c] = hitb ? bb : 0.0f;
  }
  // pixels inside the frame (a row that is, is below row_end): [half | row | pair]
  const int xa = x0 + 8 * (lane >> 5) + 2 * (lane & 3), yl = tile_y * EU4_TH + ((lane >> 2) & 7);
  float *const orow = p.out + (long long)yl * p.out_stride;
  if (C.flags & 8) eu_put<NCH>(orow, xa, qa);
  if (C.flags & 16) eu_put<NCH>(orow, xa + 1, qb);
}

// the tiles of one wave: XCD x owns the units x, x + 8, ... of EU5_UNIT_ROWS tile rows; its K waves walk
// that list in raster order, wave k taking tiles k, k + K, ...
struct eu5_iter {
  int ul, ry, rx;             // local unit, tile row inside the unit, tile column
  int du, dy, dx;             // the step of K tiles in the same terms
  int xcd, units, tiles16, tiles_y;
  __device__ __forceinline__ void start(int t0, int K)
  {
    const int per_unit = EU5_UNIT_ROWS * tiles16;
    ul = t0 / per_unit;
    ry = (t0 - ul * per_unit) / tiles16;
    rx = t0 - ul * per_unit - ry * tiles16;
    du = K / per_unit; dy = (K - du * per_unit) / tiles16; dx = K - du * per_unit - dy * tiles16;
  }
  __device__ __forceinline__ void step()
  {
    rx += dx; ry += dy; ul += du;
    if (rx >= tiles16) { rx -= tiles16; ry++; }
    if (ry >= EU5_UNIT_ROWS) { ry -= EU5_UNIT_ROWS; ul++; }
  }
  // moves on to the next tile inside the frame whose tile row has (want_plan) / has not a column plan;
  // false when the list is exhausted. plan_out: the plan of the tile row, -1: none
  __device__ __forceinline__ bool settle(const int *tileplan, bool want_plan, int &plan_out)
  {
    while (true) {
      if (ul * 8 + xcd >= units) return false;
      const int ty = (ul * 8 + xcd) * EU5_UNIT_ROWS + ry;
      if (ty < tiles_y) {
        plan_out = tileplan ? tileplan[ty] : -1;
        if ((plan_out >= 0) == want_plan) return true;
      }
      step();
    }
  }
  __device__ __forceinline__ int tile_y() const { return (ul * 8 + xcd) * EU5_UNIT_ROWS + ry; }
};

// one skewed loop over the wave's tiles with (HOIST) / without a column plan: kept apart so that each gets
// a register allocation of its own (together they spill)
template <int NCH, int DEG, int PRJ, bool HOIST, bool FAST>
__device__ __forceinline__ void eu5_loop(const eu_render_params &p, const eu4_plan &w, const float *atab, float *tile,
                                         int lane0, int wave)
{
  constexpr bool LEAN = FAST && PRJ == EU_SPHERICAL;
  const int *const tileplan = PRJ == EU_SPHERICAL ? w.tileplan : nullptr;
  // blocks are dealt round-robin to the 8 XCDs: blockIdx.x & 7 names the XCD (up to a rotation; for speed only)
  eu5_iter it;
  it.xcd = (int)(blockIdx.x & 7);
  it.tiles16 = w.tiles16; it.tiles_y = p.tiles_y;
  it.units = (p.tiles_y + EU5_UNIT_ROWS - 1) / EU5_UNIT_ROWS;
  it.start((int)(blockIdx.x >> 3) * EU5_WAVES + wave, (int)(gridDim.x >> 3) * EU5_WAVES);

  eu5_loaded Ln;
  eu5_coord<DEG, HOIST> Cn, Cc;
  eu5_plan1 Bc;
  int ty_n = 0, x0_n = 0, plan_n = -1, ty_c = 0, x0_c = 0;
  bool have_c = false, have_n = it.settle(tileplan, HOIST, plan_n);
  if (have_n) {
    ty_n = it.tile_y(); x0_n = it.rx * EU4_TW;
    eu5_load<HOIST, FAST>(p, HOIST ? w.coltab + (size_t)plan_n * p.width * EU4_COL_FLOATS : nullptr, ty_n, x0_n, lane0, Ln);
  }
#pragma unroll 1
  while (have_c || have_n) {
    // everything a stage derives from the lane index is recomputed per stage (kept live across the
    // loop it costs registers the tile code needs)
    int lane = lane0;
    asm volatile("" : "+v"(lane));
    const unsigned lds_tile = (unsigned)(unsigned long long)(eu4_lds_void)tile;
    if (have_c) eu5_boxes<DEG, HOIST, LEAN>(p, w, Cc, ty_c, x0_c, lane, Bc);
    if (have_n) eu5_settle(Ln);
    if (have_c && Bc.npass > 0) eu5_stage<NCH, DEG>(p.src, Bc.box0, lds_tile, lane);
    if (have_n) eu5_coords<NCH, DEG, PRJ, HOIST, FAST>(p, atab, Ln, ty_n, x0_n, lane, Cn);
    if (have_c && Bc.npass >= 0) eu5_finish<NCH, DEG, HOIST, FAST>(p, Cc, Bc, tile, ty_c, x0_c, lane);
    // rotate: the next tile becomes the current one; the loads of the tile after it go out
    Cc = Cn; ty_c = ty_n; x0_c = x0_n; have_c = have_n;
    if (have_n) {
      it.step();
      have_n = it.settle(tileplan, HOIST, plan_n);
      if (have_n) {
        ty_n = it.tile_y(); x0_n = it.rx * EU4_TW;
        eu5_load<HOIST, FAST>(p, HOIST ? w.coltab + (size_t)plan_n * p.width * EU4_COL_FLOATS : nullptr, ty_n, x0_n, lane, Ln);
      }
    }
  }
}

// grid: 8 * (workgroups per XCD); the launcher sizes it to what is resident at once
template <int NCH, int DEG, int PRJ, bool FAST>
__global__ __launch_bounds__(64 * EU5_WAVES, EU5_OCC) void eu_render5_kernel(const eu_render_params p, const eu4_plan w)
{
  __shared__ __attribute__((aligned(16))) float tile_all[EU5_WAVES * EU5_TEXELS * 4];
  __shared__ __attribute__((aligned(16))) float atab[768];
  const int lane0 = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  float *const tile = tile_all + wave * (EU5_TEXELS * 4);
  if (PRJ != EU_CUBEMAP && wave == 0) {
#pragma unroll
    for (int i = 0; i < 3; i++)
      __builtin_amdgcn_global_load_lds((eu4_gbl_void)(w.atab_g + i * 256 + lane0 * 4),
                                       (eu4_lds_void)(atab + i * 256), 16, 0, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  }
  __syncthreads();
  // the tile rows with a column plan (upright cubemap / rectilinear targets of a lat/lon source: the x half
  // of the coordinates is a function of the column), then the others
  if constexpr (PRJ == EU_SPHERICAL) eu5_loop<NCH, DEG, PRJ, true, FAST>(p, w, atab, tile, lane0, wave);
  eu5_loop<NCH, DEG, PRJ, false, FAST>(p, w, atab, tile, lane0, wave);
}


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
#define EU5_OCC 4           // waves per SIMD the registers are capped for (128: at 96 the tile code spills, and every
                            // scratch access costs more than the fifth wave gives: 2.06 ms with 42 of them per tile)
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
        ra = EU_MAD2(wa, ta[i].xy, ra); qa = EU_MAD2(wa, ta[i].zw, qa);
        rb = EU_MAD2(wb, tb[i].xy, rb); qb = EU_MAD2(wb, tb[i].zw, qb);
      }
      const eu_f2 ya = { wy[j].x, wy[j].x }, yb = { wy[j].y, wy[j].y };
      if (j == 0) { rga = ra * ya; bxa = qa * ya; rgb = rb * yb; bxb = qb * yb; }
      else { rga = EU_MAD2(ra, ya, rga); bxa = EU_MAD2(qa, ya, bxa); rgb = EU_MAD2(rb, yb, rgb); bxb = EU_MAD2(qb, yb, bxb); }
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
// normalisation, no bands, every ray hits, the verified constant division, brighten 1 (what a
// cubemap / rectilinear target of a full-sphere or cubemap source is). The persistent loop keeps
// every scalar it uses live: fewer of them means no SGPR spills in the tile code.
// what a 16x8 tile without a column plan reads from the stepper tables (FAST profile: 'B * c0 + A')
struct eu5_tab8 { eu_f2 c0; float A0, A1, A2, B0, B1, B2; };

template <bool FAST>
__device__ __forceinline__ void eu5_load8(const eu_render_params &p, int tile_y, int x0, int lane, eu5_tab8 &T)
{
  const int pr = lane & 3, rw = (lane >> 2) & 7, hf = lane >> 5;
  const int y = p.row_begin + tile_y * EU4_TH + rw;
  const int yc = y < p.row_end ? y : p.row_end - 1;
  const float *rt = p.row + (long long)(FAST ? yc : eu_frame_row(yc, p.band_shift, p.band_count, p.band_index)) * EU_ROW_FLOATS;
  const int xa = x0 + 8 * hf + 2 * pr, xb = xa + 1;
  const int xac = xa < p.width ? xa : p.width - 1, xbc = xb < p.width ? xb : p.width - 1;
  T.A0 = rt[0]; T.A1 = rt[1]; T.A2 = rt[2]; T.B0 = rt[3]; T.B1 = rt[4]; T.B2 = rt[5];
  T.c0 = (eu_f2){ p.col[xac], p.col[xbc] };
}

// PRE (with !HOIST, FAST): the tile's table values come in T, requested by the previous tile ahead of its
// stores, and this tile requests the next one's (have_n, ty_n, x0_n -> Tn) ahead of its own - see eu5_tile16h
template <int NCH, int DEG, int PRJ, bool HOIST, bool FAST, bool PRE = false>
__device__ __forceinline__ void eu5_tile(const eu_render_params &p, const eu4_plan &w, const float *atab,
                                         float *wtile, const float *ct, int tile_y, int x0, int lane,
                                         const eu5_tab8 *T = nullptr, bool have_n = false, int ty_n = 0, int x0_n = 0,
                                         eu5_tab8 *Tn = nullptr)
{
  static_assert(!PRE || (!HOIST && FAST), "the prefetching form exists for plain FAST tiles");
  constexpr int TEX = 4;
  constexpr int order = DEG + 1;
  const eu_src_dev &s = p.src;
#ifdef EU5_STAMPS
  unsigned long long st_[8] = { 0, 0, 0, 0, 0, 0, 0, 0 };
#endif
  EU5_STAMP(0);
  const int pr = lane & 3, rw = (lane >> 2) & 7, hf = lane >> 5;
  const int y = p.row_begin + tile_y * EU4_TH + rw;
  const bool yin = y < p.row_end;
  const int yc = yin ? y : p.row_end - 1;
  const float *rt = p.row + (long long)(FAST ? yc : eu_frame_row(yc, p.band_shift, p.band_count, p.band_index)) * EU_ROW_FLOATS;
  const int xa = x0 + 8 * hf + 2 * pr, xb = xa + 1;
  const bool va = yin && xa < p.width, vb = yin && xb < p.width;
  const int xac = xa < p.width ? xa : p.width - 1, xbc = xb < p.width ? xb : p.width - 1;

  eu_f2 tx, ty, gy;
  eu_i2 hit, ok = { -1, -1 };
  int ixa, ixb;
  eu_f2 wx[order], wy[order];
  constexpr bool LEAN = FAST && PRJ == EU_SPHERICAL;
  if constexpr (LEAN) {
    // the reference's operations in the reference's order (stepper.h ray, geometry.h:278-301,
    // environment.h:988-1006, map.h gates) in their leanest instruction forms (eu_math2.h, round 3)
    hit = (eu_i2){ -1, -1 };
    eu_i2 big0 = { 0, 0 }, big1 = { 0, 0 };
    eu_f2 lat;
    if constexpr (HOIST) {
      const float4 *ea = (const float4 *)(ct + (size_t)xac * EU4_COL_FLOATS);
      const float4 *eb = (const float4 *)(ct + (size_t)xbc * EU4_COL_FLOATS);
      const float4 a0 = ea[0], a1 = ea[1], b0 = eb[0], b1 = eb[1];
      const float A1 = rt[1], B1 = rt[4];
      const eu_f2 c0 = { p.col[xac], p.col[xbc] };
      const eu_f2 ryy = B1 * c0 + A1;
      ixa = __float_as_int(a0.x); ixb = __float_as_int(b0.x);
      tx = (eu_f2){ a0.y, b0.y };
      if constexpr (DEG >= 2) {
        wx[0] = (eu_f2){ a0.z, b0.z }; wx[1] = (eu_f2){ a0.w, b0.w }; wx[2] = (eu_f2){ a1.x, b1.x };
        if constexpr (DEG == 3) wx[3] = (eu_f2){ a1.y, b1.y };
      }
      const eu_f2 qs = { a1.z, b1.z };
      ok = (eu_i2){ (ixa != INT_MIN ? -1 : 0) & ~eu5_out_of_range3(ryy.x, qs.x, qs.x),
                    (ixb != INT_MIN ? -1 : 0) & ~eu5_out_of_range3(ryy.y, qs.y, qs.y) };
      lat = eu_atan2f_2_lean(ryy, qs, atab, 1, big0);
    } else {
      float A0, A1, A2, B0, B1, B2;
      eu_f2 c0;
      if constexpr (PRE) { A0 = T->A0; A1 = T->A1; A2 = T->A2; B0 = T->B0; B1 = T->B1; B2 = T->B2; c0 = T->c0; }
      else {
        A0 = rt[0]; A1 = rt[1]; A2 = rt[2]; B0 = rt[3]; B1 = rt[4]; B2 = rt[5];
        c0 = (eu_f2){ p.col[xac], p.col[xbc] };
      }
      const eu_f2 rx = B0 * c0 + A0, ry = B1 * c0 + A1, rz = B2 * c0 + A2;
      ok = (eu_i2){ ~eu5_out_of_range3(rx.x, ry.x, rz.x), ~eu5_out_of_range3(rx.y, ry.y, rz.y) };
      const eu_f2 q2 = rx * rx + rz * rz;
      const eu_f2 qs = eu_sqrt2_safe(q2);
      lat = eu_atan2f_2_lean(ry, qs, atab, 1, big0);
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
      ixa = (int)fx.x; ixb = (int)fx.y;
    }
    ok = ok & ~(big0 | big1);
    eu_f2 i1 = { (float)((double)lat.x - s.tex_y0), (float)((double)lat.y - s.tex_y0) };
    if (s.cdiv_ok) i1 = eu_div2_const(i1, s.ext_h, s.rcp_ext_h);
    else i1 = eu_div2_rr(i1, s.ext_h, eu_rcp_refined(s.ext_h));
    i1 = i1 * s.total_h; i1 = i1 - .5f;
    const eu_f2 sy = i1 - s.win_y_off;
    gy = eu5_gate2(sy, s.gate1, s.lower1, s.upper1);
  } else if constexpr (HOIST) {
    const float4 *ea = (const float4 *)(ct + (size_t)xac * EU4_COL_FLOATS);
    const float4 *eb = (const float4 *)(ct + (size_t)xbc * EU4_COL_FLOATS);
    const float4 a0 = ea[0], a1 = ea[1], b0 = eb[0], b1 = eb[1];
    const float A1 = rt[1], B1 = rt[4];
    const eu_f2 c0 = { p.col[xac], p.col[xbc] };
    const eu_f2 ryy = B1 * c0 + A1;
    ixa = __float_as_int(a0.x); ixb = __float_as_int(b0.x);
    ok = ok & (eu_i2){ ixa != INT_MIN ? -1 : 0, ixb != INT_MIN ? -1 : 0 };
    tx = (eu_f2){ a0.y, b0.y };
    if constexpr (DEG >= 2) {
      wx[0] = (eu_f2){ a0.z, b0.z }; wx[1] = (eu_f2){ a0.w, b0.w }; wx[2] = (eu_f2){ a1.x, b1.x };
      if constexpr (DEG == 3) wx[3] = (eu_f2){ a1.y, b1.y };
    }
    const eu_f2 qs = { a1.z, b1.z };
    const eu_f2 lat = eu_atan2f_2_tab_ok(ryy, qs, atab, 1, ok);
    hit = (eu_i2){ -1, -1 };
    if (!FAST && !s.always_hit) {
      const eu_f2 lon = { a1.w, b1.w };
      hit = (lon >= s.wex0) & (lon <= s.wex1) & (lat >= s.wex2) & (lat <= s.wex3);
    }
    eu_f2 i1 = { (float)((double)lat.x - s.tex_y0), (float)((double)lat.y - s.tex_y0) };
    if (FAST || s.cdiv_ok) i1 = eu_div2_const(i1, s.ext_h, s.rcp_ext_h);
    else i1 = i1 / s.ext_h;
    i1 = i1 * s.total_h; i1 = i1 - .5f;
    const eu_f2 sy = i1 - s.win_y_off;
    gy = eu_gate2_ok(sy, s.gate1, s.lower1, s.upper1, ok);
  } else {
    eu_ray2 r;
    {
      float A0, A1, A2, B0, B1, B2;
      eu_f2 c0;
      if constexpr (PRE) { A0 = T->A0; A1 = T->A1; A2 = T->A2; B0 = T->B0; B1 = T->B1; B2 = T->B2; c0 = T->c0; }
      else {
        A0 = rt[0]; A1 = rt[1]; A2 = rt[2]; B0 = rt[3]; B1 = rt[4]; B2 = rt[5];
        c0 = (eu_f2){ p.col[xac], p.col[xbc] };
      }
      if (!FAST && p.form == EU_FORM_BCA) {
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
      if (!FAST && p.norm_mode == EU_NORM_DIV) {
        eu_f2 sqn = r.x * r.x; sqn = sqn + r.y * r.y; sqn = sqn + r.z * r.z;
        const eu_f2 n = { sqrtf(sqn.x), sqrtf(sqn.y) };
        r.x = r.x / n; r.y = r.y / n; r.z = r.z / n;
      }
    }
    eu_f2 sx, sy;
    hit = eu_coord2_ok<PRJ, FAST>(s, r, sx, sy, atab, ok);
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
#ifdef EU5_STAMPS
  asm volatile("" : : "v"(iya), "v"(iyb), "v"(ixa), "v"(ixb));
#endif
  EU5_STAMP(2);

  // boxes of the base positions of the hitting pixels: quarters (q*, lane 15 of every row of 16
  // lanes), halves (h*, lanes 31 and 63), tile
  int q0 = INT_MAX, q1 = INT_MAX, q2 = INT_MIN, q3 = INT_MIN, h0, h1, h2, h3;
  if (hit.x) { q0 = ixa; q2 = ixa; q1 = iya; q3 = iya; }
  if (hit.y) { q0 = min(q0, ixb); q2 = max(q2, ixb); q1 = min(q1, iyb); q3 = max(q3, iyb); }
  eu5_box_reduce(q0, q1, q2, q3, h0, h1, h2, h3);
  const eu5_box full = eu5_box_join(eu5_box_at(h0, h1, h2, h3, 31), eu5_box_at(h0, h1, h2, h3, 63));
  bool clean = __ballot((hit.x && !ok.x) || (hit.y && !ok.y)) == 0ull;
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
  // passes: the tile at once, its halves or its quarters, whichever fits the slice first
  int npass = 1;
  const int f = eu5_box_fits<order>(full);
  if (f < 0) npass = 0;                               // nothing hits: zeros
  else if (f == 0) {
    npass = 2;
    if (eu5_box_fits<order>(eu5_box_at(h0, h1, h2, h3, 31)) == 0 || eu5_box_fits<order>(eu5_box_at(h0, h1, h2, h3, 63)) == 0) {
      npass = 4;
      if (eu5_box_fits<order>(eu5_box_at(q0, q1, q2, q3, 15)) == 0 || eu5_box_fits<order>(eu5_box_at(q0, q1, q2, q3, 31)) == 0 ||
          eu5_box_fits<order>(eu5_box_at(q0, q1, q2, q3, 47)) == 0 || eu5_box_fits<order>(eu5_box_at(q0, q1, q2, q3, 63)) == 0)
        npass = -1;
    }
  }
  EU5_STAMP(3);
  float *const orow = p.out + (long long)(yc - p.row_begin) * p.out_stride;
  if (npass < 0 || (npass > 0 && !clean)) {
    // not even the quarters fit (the pole of a lat/lon source, the +-180 degree seam, strong
    // minification), or a hitting pixel left the fast path of the coordinate arithmetic: left to the
    // direct-gather kernel behind this one
    if (lane == 0) {
      const int id = tile_y * w.tiles16 + x0 / EU4_TW;
      const int sh = eu4_shard_of(id);
      const int slot = atomicAdd(p.wl + EU4_WL_SHARD(sh), 1);
      p.wl[EU4_WL_ENTRIES + (size_t)slot * EU4_SHARDS + sh] = id;
    }
    if constexpr (PRE) { if (have_n) eu5_load8<FAST>(p, ty_n, x0_n, lane, *Tn); }
    return;
  }
  eu_f2 rga = { 0.0f, 0.0f }, bxa = { 0.0f, 0.0f }, rgb = { 0.0f, 0.0f }, bxb = { 0.0f, 0.0f };
  if (npass > 0) {
    const unsigned lds_tile = (unsigned)(unsigned long long)(eu4_lds_void)wtile;
    const int grp = npass == 1 ? 0 : npass == 2 ? hf : (lane >> 4);
#pragma unroll 1
    for (int pi = 0; pi < npass; pi++) {
      eu5_box bx = full;
      if (npass > 1) {
        // lane 31 / 63 of the halves' copies, lane 15 / 31 / 47 / 63 of the quarters
        const int ln = npass == 2 ? 31 + 32 * pi : 15 + 16 * pi;
        const eu5_box bh = eu5_box_at(h0, h1, h2, h3, ln), bq = eu5_box_at(q0, q1, q2, q3, ln);
        bx = npass == 2 ? bh : bq;
      }
      if (bx.mnx == INT_MAX) continue;               // a half / quarter without a hitting pixel
      const int ibw = bx.mxx - bx.mnx + order, ibh = bx.mxy - bx.mny + order;
      {
        // stage the box: lane L fetches the texel of box column L % ibw in row L / ibw of the k = 64 / ibw
        // rows ONE LDS-DMA instruction covers (an LDS-DMA instruction costs its wave 60-180 cycles of issue
        // whatever it moves: one per box row was 14-25 per tile); the LDS image is the box, rows back to back
        const unsigned tv = eu5_divtab.v[ibw];
        const int k = (int)(tv >> 24);
        const unsigned r = ((unsigned)lane * (tv & 0x1ffffu)) >> 16, c = (unsigned)lane - r * (unsigned)ibw;
        const int bx0 = bx.mnx - DEG / 2, by0 = bx.mny - DEG / 2;
        const unsigned pitchb = (unsigned)(s.es1 * 4);
        const unsigned voff = r * pitchb + c * (NCH * 4u);
        const char *sb = (const char *)(s.base + ((long long)by0 * s.es1 + (long long)bx0 * NCH));
        const unsigned long long step = (unsigned long long)k * pitchb;
        unsigned dst = lds_tile;
        const unsigned dstep = (unsigned)(k * ibw) * (TEX * 4u);
        int left = ibh;
        if ((int)r < k) {
#pragma unroll 1
          for (; left >= k; left -= k) { eu4_dma_row(dst, voff, sb); sb += step; dst += dstep; }
        }
        left = ibh % k;
        if ((int)r < left) {
          const int full = ibh / k;     // the loop above ran on other lanes only: recompute its end
          eu4_dma_row(lds_tile + (unsigned)full * dstep, voff,
                      (const char *)(s.base + ((long long)by0 * s.es1 + (long long)bx0 * NCH)) + (unsigned long long)full * step);
        }
      }
      if (pi == 0) {
        // the weights, behind the DMA issue
        if constexpr (DEG >= 2) {
          eu_weights2<DEG>(s.wm, ty, wy);
          if constexpr (!HOIST) eu_weights2<DEG>(s.wm, tx, wx);
        }
      }
      // lanes of other groups and lanes without a hit read the box origin
      const bool mine = grp == pi;
      const int oa = (mine && hit.x) ? ((iya - bx.mny) * ibw + (ixa - bx.mnx)) * TEX : 0;
      const int ob = (mine && hit.y) ? ((iyb - bx.mny) * ibw + (ixb - bx.mnx)) * TEX : 0;
      if (pi == 0) EU5_STAMP(4);
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      if (pi == 0) EU5_STAMP(5);
      if (mine) eu5_taps<NCH, DEG>((eu_lptr)wtile, oa, ob, ibw * TEX, wx, wy, tx, ty, rga, bxa, rgb, bxb);
    }
  }
#ifdef EU5_STAMPS
  asm volatile("" : : "v"(rga), "v"(bxa), "v"(rgb), "v"(bxb));
#endif
  EU5_STAMP(6);
  // environment::eval brighten (environment.h:1821-1842), zero on a miss; storer
  float qa[4] = { rga.x, rga.y, bxa.x, bxa.y }, qb[4] = { rgb.x, rgb.y, bxb.x, bxb.y };
  constexpr int ncol = (NCH == 2 || NCH == 4) ? NCH - 1 : NCH;
  const bool bright = !FAST && s.brighten != 1.0f;
#pragma unroll
  for (int c = 0; c < NCH; c++) {
    float a = qa[c], bb = qb[c];
    if (bright && c < ncol) { a = a * s.brighten; bb = bb * s.brighten; }
    qa[c] = hit.x ? a : 0.0f;
    qb[c] = hit.y ? bb : 0.0f;
  }
  if constexpr (PRE) {
    if (have_n) eu5_load8<FAST>(p, ty_n, x0_n, lane, *Tn);
    __builtin_amdgcn_sched_barrier(0);
  }
  if (va) eu_put<NCH>(orow, xa, qa);
  if (vb) eu_put<NCH>(orow, xb, qb);
  EU5_STAMP(7);
#ifdef EU5_STAMPS
  if (lane == 0 && w.stamps) {
    unsigned long long *o = w.stamps + ((size_t)tile_y * w.tiles16 + x0 / EU4_TW) * 8;
    st_[1] = (unsigned long long)(npass + 1) | (HOIST ? 16ull : 0ull);
#pragma unroll
    for (int k = 0; k < 8; k++) o[k] = st_[k];
  }
#endif
}



// stage a box: lane L fetches the texel of box column L % ibw in row L / ibw of the k = 64 / ibw rows ONE
// LDS-DMA instruction covers; the LDS image is the box, rows back to back
template <int NCH, int DEG>
__device__ __forceinline__ void eu5_stage(const eu_src_dev &s, const eu5_box &bx, unsigned lds_tile, int lane)
{
  constexpr int order = DEG + 1;
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

// A 16x16 tile on tile rows with a column plan (FAST profile, lat/lon source): FOUR pixels per lane - the
// pairs (xa, xb) of rows y and y + 8. The two pairs share everything that depends on the column (the table
// entry: base position, x weights, sqrt(rx^2 + rz^2)) and run two independent latitude chains, which is what a
// wave needs to issue a packed operation every 4 cycles instead of every 8 (a dependent chain of packed
// operations issues at half rate, and four waves per SIMD do not always have two of them in a vector phase);
// the box reduction, the DMA issue and its round trip are paid once per 256 pixels. Headline: the box of a
// 16x16 tile of an equatorial face is at most 24 x 24 texels - the slice. Returns false when the tile has to
// be rendered as two 16x8 tiles (box too large, a lane off the fast path).
// what a 16x16 tile reads from the stepper tables and the column table (22 registers)
struct eu5_tab16 {
  eu4_f4 a0, a1, b0, b1;      // the column entries of the lane's two columns
  eu_f2 c0;                   // column table of the stepper
  float A1A, B1A, A1B, B1B;   // row constants of the lane's two rows
};

__device__ __forceinline__ void eu5_load16(const eu_render_params &p, const float *ct, int tile_y, int x0, int lane, eu5_tab16 &T)
{
  const int pr = lane & 3, rw = (lane >> 2) & 7, hf = lane >> 5;
  const int yA = p.row_begin + tile_y * EU4_TH + rw, yB = yA + EU4_TH;
  const int ycA = yA < p.row_end ? yA : p.row_end - 1, ycB = yB < p.row_end ? yB : p.row_end - 1;
  const float *rtA = p.row + (long long)ycA * EU_ROW_FLOATS, *rtB = p.row + (long long)ycB * EU_ROW_FLOATS;
  const int xa = x0 + 8 * hf + 2 * pr, xb = xa + 1;
  const int xac = xa < p.width ? xa : p.width - 1, xbc = xb < p.width ? xb : p.width - 1;
  const eu4_f4 *ea = (const eu4_f4 *)(ct + (size_t)xac * EU4_COL_FLOATS);
  const eu4_f4 *eb = (const eu4_f4 *)(ct + (size_t)xbc * EU4_COL_FLOATS);
  T.a0 = ea[0]; T.a1 = ea[1]; T.b0 = eb[0]; T.b1 = eb[1];
  T.A1A = rtA[1]; T.B1A = rtA[4]; T.A1B = rtB[1]; T.B1B = rtB[4];
  T.c0 = (eu_f2){ p.col[xac], p.col[xbc] };
}

// T: the tile's table values, requested by the PREVIOUS tile ahead of its stores; this tile does the same
// for the next one (have_n, ct_n, ty_n, x0_n -> Tn). vmcnt counts in issue order: loads requested behind
// a tile's stores cannot be waited for without waiting for the stores' acknowledgement too (1-2k cycles
// at the head of every tile).
template <int NCH, int DEG>
__device__ __forceinline__ bool eu5_tile16h(const eu_render_params &p, const eu4_plan &w, const float *atab,
                                            float *wtile, const eu5_tab16 &T, int tile_y, int x0, int lane,
                                            bool have_n, const float *ct_n, int ty_n, int x0_n, eu5_tab16 &Tn)
{
  constexpr int order = DEG + 1;
  const eu_src_dev &s = p.src;
#ifdef EU5_STAMPS
  unsigned long long st_[8] = { 0, 0, 0, 0, 0, 0, 0, 0 };
#endif
  EU5_STAMP(0);
  const int pr = lane & 3, rw = (lane >> 2) & 7, hf = lane >> 5;
  const int yA = p.row_begin + tile_y * EU4_TH + rw, yB = yA + EU4_TH;
  const bool yinA = yA < p.row_end, yinB = yB < p.row_end;
  const int ycA = yinA ? yA : p.row_end - 1, ycB = yinB ? yB : p.row_end - 1;
  const int xa = x0 + 8 * hf + 2 * pr, xb = xa + 1;
  const bool vxa = xa < p.width, vxb = xb < p.width;
  const eu4_f4 a0 = T.a0, a1 = T.a1, b0 = T.b0, b1 = T.b1;
  const eu_f2 c0 = T.c0;
  const eu_f2 ryA = T.B1A * c0 + T.A1A, ryB = T.B1B * c0 + T.A1B;
  const int ixa = __float_as_int(a0.x), ixb = __float_as_int(b0.x);
  eu_f2 wx[order];
  const eu_f2 tx = { a0.y, b0.y };
  if constexpr (DEG >= 2) {
    wx[0] = (eu_f2){ a0.z, b0.z }; wx[1] = (eu_f2){ a0.w, b0.w }; wx[2] = (eu_f2){ a1.x, b1.x };
    if constexpr (DEG == 3) wx[3] = (eu_f2){ a1.y, b1.y };
  }
  const eu_f2 qs = { a1.z, b1.z };
  eu_i2 bigA = { 0, 0 }, bigB = { 0, 0 };
  // a lane is off the fast path when a column entry is, or an operand leaves the range of the FMA sequences
  int bad = ((ixa != INT_MIN && ixb != INT_MIN) ? 0 : -1) |
            eu5_out_of_range3(ryA.x, qs.x, ryB.x) | eu5_out_of_range3(ryA.y, qs.y, ryB.y);
  const eu_f2 latA = eu_atan2f_2_lean(ryA, qs, atab, 1, bigA);
  const eu_f2 latB = eu_atan2f_2_lean(ryB, qs, atab, 1, bigB);
  bad |= bigA.x | bigA.y | bigB.x | bigB.y;
  // source_t::md_to_spline, y (environment.h:988-1006)
  const float rr = s.cdiv_ok ? 0.0f : eu_rcp_refined(s.ext_h);
  eu_f2 iA = { (float)((double)latA.x - s.tex_y0), (float)((double)latA.y - s.tex_y0) };
  eu_f2 iB = { (float)((double)latB.x - s.tex_y0), (float)((double)latB.y - s.tex_y0) };
  if (s.cdiv_ok) { iA = eu_div2_const(iA, s.ext_h, s.rcp_ext_h); iB = eu_div2_const(iB, s.ext_h, s.rcp_ext_h); }
  else { iA = eu_div2_rr(iA, s.ext_h, rr); iB = eu_div2_rr(iB, s.ext_h, rr); }
  iA = iA * s.total_h; iA = iA - .5f; iB = iB * s.total_h; iB = iB - .5f;
  const eu_f2 gyA = eu5_gate2(iA - s.win_y_off, s.gate1, s.lower1, s.upper1);
  const eu_f2 gyB = eu5_gate2(iB - s.win_y_off, s.gate1, s.lower1, s.upper1);
  eu_f2 fyA, fyB;
  if constexpr (DEG & 1) { fyA = (eu_f2){ floorf(gyA.x), floorf(gyA.y) }; fyB = (eu_f2){ floorf(gyB.x), floorf(gyB.y) }; }
  else { fyA = (eu_f2){ roundf(gyA.x), roundf(gyA.y) }; fyB = (eu_f2){ roundf(gyB.x), roundf(gyB.y) }; }
  const eu_f2 tyA = gyA - fyA, tyB = gyB - fyB;
  const int iyAa = (int)fyA.x, iyAb = (int)fyA.y, iyBa = (int)fyB.x, iyBb = (int)fyB.y;
  // every ray hits (FAST): a pixel counts when it lies inside the frame
  const bool hAa = yinA && vxa, hAb = yinA && vxb, hBa = yinB && vxa, hBb = yinB && vxb;

#ifdef EU5_STAMPS
  asm volatile("" : : "v"(iyAa), "v"(iyAb), "v"(iyBa), "v"(iyBb));
#endif
  EU5_STAMP(2);
  // the tile's box
  int q0 = INT_MAX, q1 = INT_MAX, q2 = INT_MIN, q3 = INT_MIN, h0, h1, h2, h3;
  if (hAa) { q0 = ixa; q2 = ixa; q1 = iyAa; q3 = iyAa; }
  if (hAb) { q0 = min(q0, ixb); q2 = max(q2, ixb); q1 = min(q1, iyAb); q3 = max(q3, iyAb); }
  if (hBa) { q0 = min(q0, ixa); q2 = max(q2, ixa); q1 = min(q1, iyBa); q3 = max(q3, iyBa); }
  if (hBb) { q0 = min(q0, ixb); q2 = max(q2, ixb); q1 = min(q1, iyBb); q3 = max(q3, iyBb); }
  eu5_box_reduce(q0, q1, q2, q3, h0, h1, h2, h3);
  eu5_box bx = eu5_box_join(eu5_box_at(h0, h1, h2, h3, 31), eu5_box_at(h0, h1, h2, h3, 63));
  bool fast = __ballot(bad != 0 && (hAa || hAb || hBa || hBb)) == 0ull;
  if (bx.mnx != INT_MAX) {
    const int cw = (int)(s.upper0 + 0.5f), ch = (int)(s.upper1 + 0.5f);
    if (s.gate0 == 2 && bx.mnx < 0) fast = false;
    if (s.gate0 != 0 && bx.mxx >= cw - 1) fast = false;
    if (s.gate1 == 2 && bx.mny < 0) fast = false;
    if (s.gate1 != 0 && bx.mxy >= ch - 1) fast = false;
    if (eu5_box_fits<order>(bx) != 1) fast = false;
  }
  if (bx.mnx == INT_MAX || !fast) {
    if (have_n) eu5_load16(p, ct_n, ty_n, x0_n, lane, Tn);
    return bx.mnx == INT_MAX;                                // nothing inside the frame: done; else: not this way
  }
  bx.mnx = __builtin_amdgcn_readfirstlane(bx.mnx); bx.mny = __builtin_amdgcn_readfirstlane(bx.mny);
  bx.mxx = __builtin_amdgcn_readfirstlane(bx.mxx); bx.mxy = __builtin_amdgcn_readfirstlane(bx.mxy);
  const unsigned lds_tile = (unsigned)(unsigned long long)(eu4_lds_void)wtile;
  EU5_STAMP(3);
  eu5_stage<NCH, DEG>(s, bx, lds_tile, lane);
  // the y weights of both pairs, behind the DMA issue
  eu_f2 wyA[order], wyB[order];
  if constexpr (DEG >= 2) { eu_weights2<DEG>(s.wm, tyA, wyA); eu_weights2<DEG>(s.wm, tyB, wyB); }
  const int ibw = bx.mxx - bx.mnx + order;
  const int oAa = hAa ? ((iyAa - bx.mny) * ibw + (ixa - bx.mnx)) * 4 : 0, oAb = hAb ? ((iyAb - bx.mny) * ibw + (ixb - bx.mnx)) * 4 : 0;
  const int oBa = hBa ? ((iyBa - bx.mny) * ibw + (ixa - bx.mnx)) * 4 : 0, oBb = hBb ? ((iyBb - bx.mny) * ibw + (ixb - bx.mnx)) * 4 : 0;
  EU5_STAMP(4);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  EU5_STAMP(5);
  __builtin_amdgcn_sched_barrier(0);
  eu_f2 rga, bxa, rgb, bxb;
  eu5_taps<NCH, DEG>((eu_lptr)wtile, oAa, oAb, ibw * 4, wx, wyA, tx, tyA, rga, bxa, rgb, bxb);
  // (the sums are complete here, whatever the stores' conditions: keeps the evaluation out of the
  // conditional blocks and the window reads 16 bytes wide)
  asm volatile("" : "+v"(rga), "+v"(bxa), "+v"(rgb), "+v"(bxb));
  {
    float qa[4] = { rga.x, rga.y, bxa.x, bxa.y }, qb[4] = { rgb.x, rgb.y, bxb.x, bxb.y };
    float *const orow = p.out + (long long)(ycA - p.row_begin) * p.out_stride;
    if (hAa) eu_put<NCH>(orow, xa, qa);
    if (hAb) eu_put<NCH>(orow, xb, qb);
  }
  // one pair after the other: interleaved, the two sets of windows do not fit the registers
  EU5_STAMP(6);
  __builtin_amdgcn_sched_barrier(0);
  eu5_taps<NCH, DEG>((eu_lptr)wtile, oBa, oBb, ibw * 4, wx, wyB, tx, tyB, rga, bxa, rgb, bxb);
  asm volatile("" : "+v"(rga), "+v"(bxa), "+v"(rgb), "+v"(bxb));
  if (have_n) eu5_load16(p, ct_n, ty_n, x0_n, lane, Tn);
  __builtin_amdgcn_sched_barrier(0);
  {
    float qa[4] = { rga.x, rga.y, bxa.x, bxa.y }, qb[4] = { rgb.x, rgb.y, bxb.x, bxb.y };
    float *const orow = p.out + (long long)(ycB - p.row_begin) * p.out_stride;
    if (hBa) eu_put<NCH>(orow, xa, qa);
    if (hBb) eu_put<NCH>(orow, xb, qb);
  }
  EU5_STAMP(7);
#ifdef EU5_STAMPS
  if (lane == 0 && w.stamps) {
    unsigned long long *o = w.stamps + ((size_t)tile_y * w.tiles16 + x0 / EU4_TW) * 8;
    st_[1] = 6ull | 16ull;      // class: hoist 1, "npass 5" = a 16x16 tile (taps column: pair A only, store column: pair B + stores)
#pragma unroll
    for (int k = 0; k < 8; k++) o[k] = st_[k];
  }
#endif
  return true;
}

// the tiles of one wave: XCD x owns the units x, x + 8, ... of UNIT tile rows (of RPT 16x8 tile rows each);
// its K waves walk that list in raster order, wave k taking tiles k, k + K, ...
template <int UNIT>
struct eu5_iter {
  int ul, ry, rx;             // local unit, tile row inside the unit, tile column
  int du, dy, dx;             // the step of K tiles in the same terms
  int xcd, units, tiles16;
  __device__ __forceinline__ void start(int t0, int K)
  {
    const int per_unit = UNIT * tiles16;
    ul = t0 / per_unit;
    ry = (t0 - ul * per_unit) / tiles16;
    rx = t0 - ul * per_unit - ry * tiles16;
    du = K / per_unit; dy = (K - du * per_unit) / tiles16; dx = K - du * per_unit - dy * tiles16;
  }
  __device__ __forceinline__ void step()
  {
    rx += dx; ry += dy; ul += du;
    if (rx >= tiles16) { rx -= tiles16; ry++; }
    if (ry >= UNIT) { ry -= UNIT; ul++; }
  }
  __device__ __forceinline__ bool done() const { return ul * 8 + xcd >= units; }
  __device__ __forceinline__ int row() const { return (ul * 8 + xcd) * UNIT + ry; }
  // The tile column of the current position: the list's column rotated by an amount that depends on the row.
  // With K waves per XCD and K a multiple of half the tiles per row (640 = 2.5 x 256 for the headline) wave k
  // would otherwise render the SAME two tile columns of every row - and the columns through a pole cost two to
  // four passes per tile, the columns at the face's edge one: the launch then ends with the waves that own
  // the pole columns (measured: the tiles' own cycles add up to 0.79 ms of a 1.12 ms launch).
  __device__ __forceinline__ int col() const
  {
    const unsigned h = ((unsigned)row() * 0x9E3779B1u) >> 16;
    int c = rx + (int)((h * (unsigned)tiles16) >> 16);
    if (c >= tiles16) c -= tiles16;
    return c;
  }
};

// tile rows 2m and 2m + 1 form a 16x16 row of the first loop when both have the same column plan
// (the plan table is read through the scalar cache: as a vector load its wait - vmcnt counts in issue order -
// is also a wait for the stores of the tile just finished, 1-2k cycles per iteration)
typedef const __attribute__((address_space(4))) int *eu5_cint;
__device__ __forceinline__ int eu5_pair_plan(const int *tileplan, int tiles_y, int m)
{
  if (2 * m + 1 >= tiles_y) return -1;
  const eu5_cint tp = (eu5_cint)tileplan;
  const int a = tp[2 * m], b = tp[2 * m + 1];
  return a == b ? a : -1;
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
#ifdef EU5_STAMPS
  const unsigned long long ws0 = __builtin_amdgcn_s_memtime();
  unsigned long long ws1 = ws0;
#endif
  // blocks are dealt round-robin to the 8 XCDs: blockIdx.x & 7 names the XCD (up to a rotation;
  // for speed only)
  const int xcd = (int)(blockIdx.x & 7);
  const int K = (int)(gridDim.x >> 3) * EU5_WAVES;
  const int t0 = (int)(blockIdx.x >> 3) * EU5_WAVES + wave;
  constexpr bool PAIRS = FAST && PRJ == EU_SPHERICAL;
  // In both loops the plan of the NEXT position is requested before the current tile is rendered: a scalar
  // load from global memory takes 500+ cycles under this load, and a wave has nothing else to do while it
  // waits for the plan of the tile it is about to start (measured: 1.6-2.3k cycles per iteration outside
  // the tiles, a quarter of a wave's life).
  if constexpr (PAIRS) {
    // first the pairs of tile rows with a common column plan, as 16x16 tiles: position i of the XCD's list
    // (w.l1_rows: built on the host with the plans, so that no position is looked at only to be skipped) is
    // column i % tiles16 - rotated by the row, see eu5_iter::col - of double row l1_rows[i / tiles16]
    const int n1 = (w.l1_off[xcd + 1] - w.l1_off[xcd]) * w.tiles16;
    const eu5_cint rows1 = (eu5_cint)w.l1_rows + 2 * w.l1_off[xcd];
    int pos1 = t0;
    auto seek = [&](int &m, int &tcol, int &plan) -> bool {
      if (pos1 >= n1) return false;
      const int r = (int)(((unsigned long long)(unsigned)pos1 * w.l1_magic) >> 40);     // pos1 / tiles16
      m = rows1[2 * r]; plan = rows1[2 * r + 1];
      const unsigned h = ((unsigned)m * 0x9E3779B1u) >> 16;
      int c = pos1 - r * w.tiles16 + (int)((h * (unsigned)w.tiles16) >> 16);
      if (c >= w.tiles16) c -= w.tiles16;
      tcol = c;
      pos1 += K;
      return true;
    };
    // two copies of the body, the table values alternating between Ta and Tb: handing them from "next" to
    // "current" by assignment is a use, and a use is a wait for the loads that have just been requested
    int m_a = 0, col_a = 0, plan_a = -1, m_b = 0, col_b = 0, plan_b = -1;
    bool have_a = seek(m_a, col_a, plan_a), have_b = false;
    eu5_tab16 Ta, Tb;
    if (have_a) eu5_load16(p, w.coltab + (size_t)plan_a * p.width * EU4_COL_FLOATS, 2 * m_a, col_a * EU4_TW, lane0, Ta);
    auto fallback = [&](int m, int tcol, int lane) {
      // both 16x8 tiles to the direct-gather kernel (a rare event: the +-180 degree seam, a box beyond the slice)
      if (lane < 2) {
        const int id = (2 * m + lane) * w.tiles16 + tcol;
        const int sh = eu4_shard_of(id);
        const int slot = atomicAdd(p.wl + EU4_WL_SHARD(sh), 1);
        p.wl[EU4_WL_ENTRIES + (size_t)slot * EU4_SHARDS + sh] = id;
      }
    };
#pragma unroll 1
    while (have_a) {
      {
        have_b = seek(m_b, col_b, plan_b);
        int lane = lane0;
        asm volatile("" : "+v"(lane));
        const float *ct_n = w.coltab + (size_t)(have_b ? plan_b : 0) * p.width * EU4_COL_FLOATS;
        if (!eu5_tile16h<NCH, DEG>(p, w, atab, tile, Ta, 2 * m_a, col_a * EU4_TW, lane, have_b, ct_n, 2 * m_b, col_b * EU4_TW, Tb))
          fallback(m_a, col_a, lane);
      }
      if (!have_b) break;
      {
        have_a = seek(m_a, col_a, plan_a);
        int lane = lane0;
        asm volatile("" : "+v"(lane));
        const float *ct_n = w.coltab + (size_t)(have_a ? plan_a : 0) * p.width * EU4_COL_FLOATS;
        if (!eu5_tile16h<NCH, DEG>(p, w, atab, tile, Tb, 2 * m_b, col_b * EU4_TW, lane, have_a, ct_n, 2 * m_a, col_a * EU4_TW, Ta))
          fallback(m_b, col_b, lane);
      }
    }
  }
#ifdef EU5_STAMPS
  ws1 = __builtin_amdgcn_s_memtime();
#endif
  // XCD x owns the units x, x + 8, ... of EU5_UNIT_ROWS tile rows; its waves walk that list in raster
  // order, wave k taking tiles k, k + K, ...
  eu5_iter<EU5_UNIT_ROWS> it;
  it.xcd = xcd; it.tiles16 = w.tiles16;
  it.units = (p.tiles_y + EU5_UNIT_ROWS - 1) / EU5_UNIT_ROWS;
  it.start(t0, K);
  if constexpr (PAIRS) {
    // The rows the first loop left: 16x8 tiles without a column plan (a row whose partner has another plan does
    // without its own here). Their cost varies - one, two or four passes, a hand-over to the work list - so they
    // are dealt out DYNAMICALLY: a wave takes the next batch of two neighbouring tiles from its XCD's queue
    // (w.l2_rows; one returning atomic per batch, requested a tile before its answer is needed). Dealt statically
    // the waves of a launch finished between 1.77 and 2.46 M cycles (average 2.10 M): a sixth of the launch was
    // waiting for the waves that had drawn the poles. Each tile requests the next one's table values ahead of its
    // stores, as in the first loop.
    int *const queue = p.wl + EU4_WL_DYN(xcd);
    const int nbatch = (w.l2_off[xcd + 1] - w.l2_off[xcd]) * w.l2_half;
    const eu5_cint rows2 = (eu5_cint)w.l2_rows + w.l2_off[xcd];
    auto take = [&]() -> int {                      // the queue's next batch (lane 0's value counts)
      int v = 0;
      if (lane0 == 0) v = atomicAdd(queue, 1);
      return v;                                     // read with readfirstlane where it is needed, not here
    };
    auto decode = [&](int j, int &ty, int &tcol) -> bool {
      if (j >= nbatch) return false;
      const int r = (int)(((unsigned long long)(unsigned)j * w.l2_magic) >> 40);      // j / l2_half
      ty = rows2[r]; tcol = 2 * (j - r * w.l2_half);
      return true;
    };
    int ty_a = 0, col_a = 0, ty_n = 0, col_n = 0;
    bool have = decode(__builtin_amdgcn_readfirstlane(take()), ty_a, col_a);
    eu5_tab8 Ta, Tb;
    if (have) eu5_load8<FAST>(p, ty_a, col_a * EU4_TW, lane0, Ta);
#pragma unroll 1
    while (have) {
      const int jn_lane = take();                   // in flight across the first tile of this batch
      const bool second = col_a + 1 < w.tiles16;    // (an odd number of tiles per row: the last batch is one tile)
      {
        int lane = lane0;
        asm volatile("" : "+v"(lane));
        eu5_tile<NCH, DEG, PRJ, false, FAST, true>(p, w, atab, tile, nullptr, ty_a, col_a * EU4_TW, lane, &Ta, second, ty_a,
                                                   (col_a + 1) * EU4_TW, &Tb);
      }
      const bool have_n = decode(__builtin_amdgcn_readfirstlane(jn_lane), ty_n, col_n);
      if (second) {
        int lane = lane0;
        asm volatile("" : "+v"(lane));
        eu5_tile<NCH, DEG, PRJ, false, FAST, true>(p, w, atab, tile, nullptr, ty_a, (col_a + 1) * EU4_TW, lane, &Tb, have_n, ty_n,
                                                   col_n * EU4_TW, &Ta);
      } else if (have_n) eu5_load8<FAST>(p, ty_n, col_n * EU4_TW, lane0, Ta);
      ty_a = ty_n; col_a = col_n; have = have_n;
    }
  } else {
    int plan_n = (it.done() || it.row() >= p.tiles_y) ? -2 : (PRJ == EU_SPHERICAL ? ((eu5_cint)w.tileplan)[it.row()] : -1);
#pragma unroll 1
    while (!it.done()) {
      const int tile_y = it.row(), tcol = it.col(), plan = plan_n;
      it.step();
      plan_n = (it.done() || it.row() >= p.tiles_y) ? -2 : (PRJ == EU_SPHERICAL ? ((eu5_cint)w.tileplan)[it.row()] : -1);
      if (plan == -2) continue;
      // everything a tile derives from the lane index is recomputed per tile (kept live across the
      // loop it costs registers the tile code needs)
      int lane = lane0;
      asm volatile("" : "+v"(lane));
      if (plan >= 0)
        eu5_tile<NCH, DEG, PRJ, PRJ == EU_SPHERICAL, FAST>(p, w, atab, tile, w.coltab + (size_t)plan * p.width * EU4_COL_FLOATS,
                                                            tile_y, tcol * EU4_TW, lane);
      else
        eu5_tile<NCH, DEG, PRJ, false, FAST>(p, w, atab, tile, nullptr, tile_y, tcol * EU4_TW, lane);
    }
  }
#ifdef EU5_STAMPS
  if (lane0 == 0 && w.stamps) {
    unsigned long long *o = w.stamps + (size_t)w.tiles16 * p.tiles_y * 8 + ((size_t)blockIdx.x * EU5_WAVES + wave) * 4;
    o[0] = ws0; o[1] = ws1; o[2] = __builtin_amdgcn_s_memtime(); o[3] = 1;
  }
#endif
}


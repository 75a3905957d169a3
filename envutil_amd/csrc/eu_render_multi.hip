// Multi-facet rendering: N steppers (one per source facet, each with that
// facet's composed rotation) feed a synopsis that picks / composites the facets
// per pixel.
//   fusion_t                zimt/get.h:1181-1242
//   synopsis_t (twining)    envutil_payload.cc:587-691
//   _voronoi_syn            envutil_payload.cc:762-957   (1 or 3 channels)
//   _voronoi_syn_plus       envutil_payload.cc:964-1233  (alpha: 2 or 4 channels)
//   _hdr_merge_syn          envutil_payload.cc:1325-1626 (args.synopsis == "hdr_merge")
//
// One thread per output pixel, 64x4 tiles like eu_render_kernel. The reference
// takes three decisions per 16-lane VECTOR (which facets enter the layer list,
// "one facet on top everywhere", "fully opaque"); a wavefront holds four such
// vectors (its 64 pixels start at a multiple of 64 inside a 512-pixel segment),
// so those decisions are wavefront ballots masked to 16-lane groups - the
// reference's any_of/all_of, bit for bit.
//
// The facets are walked by run-time loops (any count up to 64): the facet index
// is wave-uniform, its parameters come through the scalar cache. The mask pass
// computes every facet's source coordinate once; the coordinates (and, with
// alpha, the z scores the layers are sorted by) of up to 16 facets stay in LDS,
// one slot per thread, and the evaluation of the winning facet(s) starts from
// them. Facets whose evaluation differs between lanes are handled by a waterfall
// loop (readlane makes the index uniform, the lanes that want it go together).
#include <cstring>
#include "eu_render_dev.h"

#define EU_MULTI_MAXF 64     // facets per job the mask-based alpha compositing takes (one bit each); beyond: eu_synopsis_big
#define EU_MULTI_KEEP 16     // facets whose coordinates are kept in LDS (3 KB each per workgroup)

struct eu_multi_params {
  int width, height, row_begin, row_end;
  int form, norm_mode, twine, ntaps, nch, nfct, plus;
  const float *col;          // [4][width], shared by all facets
  const float *row;          // [nfct][height][EU_ROW_FLOATS]
  const float *taps;         // [ntaps][3], x/y scaled by 4
  const eu_src_dev *srcs;    // [nfct]
  float *out;
  long long out_stride;
  int tiles_x, tiles_y;
  int band_shift, band_count, band_index;   // eu_frame_row
  int hdr, hdr_low, hdr_high;               // _hdr_merge_syn: the facets that rule the shadows / the highlights
  const eu_generic *gen;                    // [nfct] or nullptr: facets stepped by generic_stepper (translation)
  eu_inv_planar inv;                        // tf22 of a --single job
  const float *rej;                         // [nfct][EU_REJ_STRIDE] or nullptr: early-miss tables (eu_multi_maybe)
};

// Early miss, second stage (round 3). The exact hit test of a facet costs ~300 vector instructions (two atan2f,
// sincosf in double, the lens polynomial, md_to_spline) and config 5 ran it 3.1 times per pixel: for every
// facet whose CORNER cone (rej_cos) holds the ray. The window is a square, two thirds of that cone. For a
// fisheye facet the planar coordinate is c = R(theta) * (rx, ry) / |(rx, ry)| + shift with R = theta * lens
// polynomial, and theta is a function of u = rz / |ray|: a table of a LOWER bound of R over bins of u (built
// on the host in double, 0.2 % below the smallest value of the bin and its neighbours) turns "c.x beyond the
// right edge" into a few approximate operations (rsq, one table read). Conservative by construction: a ray is
// dropped only when its coordinate lies beyond an edge moved OUT by 0.1 % of the window, far more than float
// rounding moves it; everything else takes the exact test as before. Header of a facet's table:
// [0] u0, [1] bins per unit of u, [2] 0 = no table, [4] [5] shift, [6]..[9] the edges x0 x1 y0 y1 moved out.
#define EU_REJ_N 1024
#define EU_REJ_HDR 16
#define EU_REJ_STRIDE (EU_REJ_HDR + EU_REJ_N)
__device__ __forceinline__ bool eu_multi_maybe(const eu_multi_params &p, int f, const eu_src_dev &s, float rx, float ry, float rz)
{
  const float n2 = rx * rx + ry * ry, n3 = n2 + rz * rz;
  // stage one: the whole window lies inside a cone around the facet's axis
  bool maybe = !(rz < s.rej_cos * __builtin_amdgcn_sqrtf(n3));
  if (p.rej) {
    const float *tb = p.rej + (size_t)f * EU_REJ_STRIDE;         // f is wave-uniform: scalar loads
    if (tb[2] == 2.0f) {
      // no table: theta = acos(u) >= sqrt(2 t) (1 + t / 12 + 3 t^2 / 160), t = 1 - u (the series of acos in
      // sqrt(2 t), every term positive: cut off it is a lower bound, 0.13 % low at 65 degrees, 0.9 % at 92), and R is
      // increasing in theta (checked on the host), so R(bound) <= R(theta). Nothing is read per lane.
      const float t = 1.0f - rz * __builtin_amdgcn_rsqf(n3);
      const float th = __builtin_amdgcn_sqrtf(2.0f * t) * (1.0f + t * (0.083333f + t * 0.01875f));
      const float x = th * tb[10];
      const float lo = th * (tb[11] + x * (tb[12] + x * (tb[13] + x * tb[14])));          // 0.2 % folded into tb[11..14]
      const float ir = __builtin_amdgcn_rsqf(n2);
      const float a0 = lo * (rx * ir) + tb[4], a1 = lo * (ry * ir) + tb[5];
      const bool out = (rx >= 0.0f ? a0 > tb[7] : a0 < tb[6]) || (ry >= 0.0f ? a1 > tb[9] : a1 < tb[8]);
      maybe = maybe && !(out && t > 0.0f);
    } else if (tb[2] != 0.0f) {
      const float u = rz * __builtin_amdgcn_rsqf(n3);
      int k = (int)((u - tb[0]) * tb[1]);
      k = min(max(k, 0), EU_REJ_N - 1);
      const float lo = tb[EU_REJ_HDR + k];
      const float ir = __builtin_amdgcn_rsqf(n2);                 // (a ray on the axis: NaN below, no early miss)
      const float a0 = lo * (rx * ir) + tb[4], a1 = lo * (ry * ir) + tb[5];
      const bool out = (rx >= 0.0f ? a0 > tb[7] : a0 < tb[6]) || (ry >= 0.0f ? a1 > tb[9] : a1 < tb[8]);
      maybe = maybe && !out;
    }
  }
  return maybe;
}

struct eu_pix { int x, y; };

// ray of facet f for this pixel; variant 0: r00, 1: x-biased, 2: y-biased
template <bool GEN>
__device__ __forceinline__ void eu_multi_ray(const eu_multi_params &p, int f, int variant,
                                             const eu_pix &px, float &rx, float &ry, float &rz)
{
  const float *rowt = p.row + ((long long)f * p.height + eu_frame_row(px.y, p.band_shift, p.band_count, p.band_index)) * EU_ROW_FLOATS
                      + (variant == 2 ? EU_ROW_VARIANT : 0);
  const float *ca = variant == 1 ? p.col + 2 * p.width : p.col;
  if constexpr (GEN) {
    if (p.gen && p.gen[f].on) {             // f is wave-uniform
      // generic_stepper<float, LANES, true>: the ray is normalised (stepper.h:431-434)
      eu_stepper<true>(EU_FORM_GENERIC, EU_NORM_DIV, ca, ca, rowt, px.x, rx, ry, rz, &p.gen[f],
                       p.col + (variant == 1 ? 5 : 4) * (long long)p.width,
                       (p.inv.shear | p.inv.shift | p.inv.lcp) ? &p.inv : nullptr);
      return;
    }
  }
  eu_stepper<false>(p.form, p.norm_mode, ca, ca + p.width, rowt, px.x, rx, ry, rz);
}

// the ray the synopsis sees for facet f: the stepper's, or the twining tap's
// p0 + cx * du + cy * dv (payload.cc:669-675)
template <bool GEN>
__device__ __forceinline__ void eu_syn_ray(const eu_multi_params &p, int f, const eu_pix &px,
                                           bool tap, float cx, float cy, float &rx, float &ry,
                                           float &rz)
{
  eu_multi_ray<GEN>(p, f, 0, px, rx, ry, rz);
  if (tap) {
    float ax, ay, az, bx, by, bz;
    eu_multi_ray<GEN>(p, f, 1, px, ax, ay, az);
    eu_multi_ray<GEN>(p, f, 2, px, bx, by, bz);
    float dux = ax - rx, duy = ay - ry, duz = az - rz;
    float dvx = bx - rx, dvy = by - ry, dvz = bz - rz;
    rx = rx + cx * dux + cy * dvx;
    ry = ry + cx * duy + cy * dvy;
    rz = rz + cx * duz + cy * dvz;
  }
}

#ifdef EU_MULTI_NCH
#ifdef EU_MULTI_STAMPS
// diagnostic build (tools/multi_stamps.py): shader-clock cycles per phase of eu_synopsis's alpha path, summed
// over the waves of a launch: [0] mask pass, [1] the top facet / all-top evaluation, [2] compositing, [3] waves
__device__ unsigned long long eu_multi_stamp_acc[1024 * 4];      // sharded by workgroup: one hot address serialises the launch
#define EU_MST(k) do { asm volatile("" ::: "memory"); mst_[k] = __builtin_amdgcn_s_memtime(); asm volatile("" ::: "memory"); } while (0)
#else
#define EU_MST(k) do { } while (0)
#endif
// per-thread slots in dynamic LDS: [z | sx | sy][facet][256 threads]
struct eu_slots {
  float *z, *sx, *sy;        // this thread's slot of facet 0; facets are 256 floats apart
  bool keep;                 // coordinates are stored (nfct <= EU_MULTI_KEEP)
};

// the facet's environment with channel adaption; a real call (one body per
// source channel count and degree, shared by all kernels of this file)
template <int SN, int DEG>
__device__ __noinline__ float4 eu_env_adapted(const eu_src_dev *s, int out_n, bool hit, float sx,
                                              float sy)
{
  float t[4] = { 0.0f, 0.0f, 0.0f, 0.0f };
  eu_environment_repix_at<SN, DEG>(*s, out_n, hit, sx, sy, t);
  return make_float4(t[0], t[1], t[2], t[3]);
}

// facet f (wave-uniform) at this lane's source coordinate
template <int NCH, int DEG>
__device__ __forceinline__ void eu_env_facet(const eu_src_dev &s, bool hit, float sx, float sy,
                                             float *out)
{
  if (s.nch == NCH) {
    eu_environment_at<NCH, DEG>(s, hit, sx, sy, out);
  } else {
    // a facet with another channel count: repix_t inside its environment
    // object (environment.h:1846-1900); f is wave-uniform, so is this switch
    float4 t;
    switch (s.nch) {
      case 1: t = eu_env_adapted<1, DEG>(&s, NCH, hit, sx, sy); break;
      case 2: t = eu_env_adapted<2, DEG>(&s, NCH, hit, sx, sy); break;
      case 3: t = eu_env_adapted<3, DEG>(&s, NCH, hit, sx, sy); break;
      default: t = eu_env_adapted<4, DEG>(&s, NCH, hit, sx, sy); break;
    }
    const float tt[4] = { t.x, t.y, t.z, t.w };
#pragma unroll
    for (int c = 0; c < NCH; c++) out[c] = tt[c];
  }
}

// evaluate facet `want` (wave-divergent, -1: none) for this lane; hitm: the
// facets this lane's ray hits (bit per facet)
template <int NCH, int DEG, bool GEN>
__device__ __forceinline__ void eu_eval_facet(const eu_multi_params &p, int want, const eu_pix &px,
                                              bool tap, float cx, float cy, const eu_slots &sl,
                                              unsigned long long hitm, float *out)
{
#pragma unroll
  for (int c = 0; c < NCH; c++) out[c] = 0.0f;
  int pending = want;
  while (true) {
    unsigned long long m = __ballot(pending >= 0);
    if (!m) break;
    int first = __ffsll((long long)m) - 1;
    int f = __builtin_amdgcn_readlane(pending, first);
    if (pending == f) {
      const eu_src_dev &s = p.srcs[f];
      float sx, sy;
      bool hit;
      if (sl.keep && !s.mask_all) {
        sx = sl.sx[f * 256]; sy = sl.sy[f * 256];
        hit = (hitm >> f) & 1ull;
      } else {
        float rx, ry, rz;
        int face;
        eu_syn_ray<GEN>(p, f, px, tap, cx, cy, rx, ry, rz);
        hit = eu_source_coordinate(s, rx, ry, rz, sx, sy, face);
      }
      eu_env_facet<NCH, DEG>(s, hit, sx, sy, out);
      pending = -1;
    }
  }
}

// one synopsis evaluation for this lane
template <int NCH, int DEG, bool PLUS, bool GEN>
__device__ __forceinline__ void eu_synopsis(const eu_multi_params &p, const eu_pix &px,
                                            bool live, bool tap, float cx, float cy,
                                            const eu_slots &sl, float *out)
{
  const int nf = p.nfct;
  if constexpr (!PLUS) {
    // _voronoi_syn: get_mask + z score of every facet; the largest z wins,
    // strict '>' keeps the earlier facet. The champion's coordinate is kept.
    int champ = -1;
    float max_z = -3.402823466e+38f;          // numeric_limits<float>::lowest()
    float csx = 0.0f, csy = 0.0f;
    bool have = false;
#pragma unroll 1
    for (int f = 0; f < nf; f++) {
      float rx, ry, rz, sx = 0.0f, sy = 0.0f;
      int face;
      eu_syn_ray<GEN>(p, f, px, tap, cx, cy, rx, ry, rz);
      const eu_src_dev &s = p.srcs[f];
      const bool masked = !s.mask_all;        // wave-uniform
      bool hit = true;
      if (masked) {
        // whole wavefront provably outside the facet's window: skip the exact test
        const bool maybe = eu_multi_maybe(p, f, s, rx, ry, rz);
        hit = false;
        if (__ballot(maybe)) hit = eu_source_coordinate(s, rx, ry, rz, sx, sy, face);
      }
      const float z = rz * s.recip_step;
      if (hit && live && (f == 0 || z > max_z)) {
        // f == 0: the reference seeds champion and max_z with facet 0 where it is valid
        champ = f; max_z = z; csx = sx; csy = sy; have = masked;
      }
    }
    // evaluate the champion: waterfall over the facets the lanes chose
#pragma unroll
    for (int c = 0; c < NCH; c++) out[c] = 0.0f;
    int pending = champ;
    while (true) {
      unsigned long long m = __ballot(pending >= 0);
      if (!m) break;
      int first = __ffsll((long long)m) - 1;
      int f = __builtin_amdgcn_readlane(pending, first);
      if (pending == f) {
        const eu_src_dev &s = p.srcs[f];
        float sx = csx, sy = csy;
        bool hit = true;
        if (!have) {
          float rx, ry, rz;
          int face;
          eu_syn_ray<GEN>(p, f, px, tap, cx, cy, rx, ry, rz);
          hit = eu_source_coordinate(s, rx, ry, rz, sx, sy, face);
        }
        eu_env_facet<NCH, DEG>(s, hit, sx, sy, out);
        pending = -1;
      }
    }
    return;
  } else {
    const int lane = threadIdx.x & 63;
    const int grp = lane >> 4;
#ifdef EU_MULTI_STAMPS
    unsigned long long mst_[4];
    int nexact_ = 0;
#endif
    EU_MST(0);
    const unsigned long long live_m = __ballot(live);
    const unsigned live_g = (unsigned)(live_m >> (16 * grp)) & 0xffffu;
    unsigned long long valid = 0, hitm = 0;
    // next_best of this lane's vector: the last facet valid for any of its lanes
    int next_best = -1;
#pragma unroll 1
    for (int f = 0; f < nf; f++) {
      float rx, ry, rz, sx = 0.0f, sy = 0.0f;
      int face;
      eu_syn_ray<GEN>(p, f, px, tap, cx, cy, rx, ry, rz);
      const eu_src_dev &s = p.srcs[f];
      bool hit = true;
      if (!s.mask_all) {
        // whole wavefront provably outside the facet's window: skip the exact test
        const bool maybe = eu_multi_maybe(p, f, s, rx, ry, rz);
        hit = false;
        if (__ballot(maybe)) {
          hit = eu_source_coordinate(s, rx, ry, rz, sx, sy, face);
#ifdef EU_MULTI_STAMPS
          nexact_++;
#endif
        }
      }
      sl.z[f * 256] = rz * s.recip_step;
      if (sl.keep) { sl.sx[f * 256] = sx; sl.sy[f * 256] = sy; }
      if (hit) hitm |= 1ull << f;
      const bool v = hit && live;
      if (v) valid |= 1ull << f;
      const unsigned long long bm = __ballot(v);
      if ((unsigned)(bm >> (16 * grp)) & 0xffffu) next_best = f;
    }
    // the lane's nearest valid facet that is not used yet
    auto pick = [&](unsigned long long used) {
      int best = -1;
      float bz = 0.0f;
      unsigned long long todo = valid & ~used;
      while (todo) {
        const int f = __ffsll((long long)todo) - 1;
        todo &= todo - 1;
        const float z = sl.z[f * 256];
        if (best < 0 || z > bz) { best = f; bz = z; }
      }
      return best;
    };
    EU_MST(1);
    const int top = pick(0ull);
#pragma unroll
    for (int c = 0; c < NCH; c++) out[c] = 0.0f;
    bool done = !live;
    if (next_best < 0) done = true;           // layers == 0 for this vector
    // "one facet on top of the whole vector" + "opaque everywhere": take it as is
    {
      unsigned long long tm = __ballot(live && top == next_best);
      bool all_top = !done && ((unsigned)(tm >> (16 * grp)) & 0xffffu) == live_g;
      float help[NCH];
      eu_eval_facet<NCH, DEG, GEN>(p, all_top ? next_best : -1, px, tap, cx, cy, sl, hitm, help);
      unsigned long long om = __ballot(all_top && help[NCH - 1] >= 1.0f);
      bool opaque = all_top && ((unsigned)(om >> (16 * grp)) & 0xffffu) == live_g;
      if (opaque) {
#pragma unroll
        for (int c = 0; c < NCH; c++) out[c] = help[c];
        done = true;
      }
    }
    EU_MST(2);
    // general path: composite the lane's valid facets, nearest first
    unsigned long long used = 0;
    int layer = 0;
    while (true) {
      int f = done ? -1 : pick(used);
      if (!__ballot(f >= 0)) break;
      float help[NCH];
      eu_eval_facet<NCH, DEG, GEN>(p, f, px, tap, cx, cy, sl, hitm, help);
      if (f >= 0) {
        used |= 1ull << f;
        if (layer == 0) {
#pragma unroll
          for (int c = 0; c < NCH; c++) out[c] = help[c];
        } else {
          const float a = out[NCH - 1];
#pragma unroll
          for (int c = 0; c < NCH; c++) out[c] = out[c] + (1.0f - a) * help[c];
        }
        layer++;
      }
    }
#ifdef EU_MULTI_STAMPS
    EU_MST(3);
    if (lane == 0) {
      unsigned long long *acc = eu_multi_stamp_acc + 4 * (blockIdx.x & 1023);
      atomicAdd(&acc[0], mst_[1] - mst_[0]);
      atomicAdd(&acc[1], mst_[2] - mst_[1]);
      atomicAdd(&acc[2], mst_[3] - mst_[2]);
      atomicAdd(&acc[3], 1ull + ((unsigned long long)nexact_ << 32));      // waves, exact hit tests (wave-level) above bit 32
    }
#endif
  }
}

// _voronoi_syn_plus for MORE facets than there are mask bits: nothing is kept per facet. A pass over all
// facets finds this lane's nearest valid facet BEHIND the layer composited last - (z, facet) smaller in
// the order the reference's layer list has (z descending, the earlier facet first among equals) - by
// recomputing every facet's ray, hit test and z score; one such pass per layer. Slow (facets x layers)
// and without limit; jobs of up to 64 facets use eu_synopsis.
template <int NCH, int DEG, bool GEN>
__device__ __forceinline__ void eu_synopsis_big(const eu_multi_params &p, const eu_pix &px, bool live,
                                                bool tap, float cx, float cy, const eu_slots &sl, float *out)
{
  const int nf = p.nfct;
  const int grp = (threadIdx.x & 63) >> 4;
  const unsigned live_g = (unsigned)(__ballot(live) >> (16 * grp)) & 0xffffu;
  // the nearest valid facet behind (lz, lf); first = true: the nearest of all. Also next_best of the
  // lane's vector: the last facet valid for any of its lanes
  int next_best = -1;
  auto pick = [&](bool first, float lz, int lf, float &bz) {
    int best = -1;
    bz = 0.0f;
#pragma unroll 1
    for (int f = 0; f < nf; f++) {
      float rx, ry, rz, sx = 0.0f, sy = 0.0f;
      int face;
      eu_syn_ray<GEN>(p, f, px, tap, cx, cy, rx, ry, rz);
      const eu_src_dev &s = p.srcs[f];
      bool hit = true;
      if (!s.mask_all) {
        const bool maybe = eu_multi_maybe(p, f, s, rx, ry, rz);
        hit = false;
        if (__ballot(maybe)) hit = eu_source_coordinate(s, rx, ry, rz, sx, sy, face);
      }
      const bool v = hit && live;
      if (first) {
        const unsigned long long bm = __ballot(v);
        if ((unsigned)(bm >> (16 * grp)) & 0xffffu) next_best = f;
      }
      const float z = rz * s.recip_step;
      const bool behind = first || z < lz || (z == lz && f > lf);
      if (v && behind && (best < 0 || z > bz)) { best = f; bz = z; }
    }
    return best;
  };
  float tz;
  const int top = pick(true, 0.0f, -1, tz);
#pragma unroll
  for (int c = 0; c < NCH; c++) out[c] = 0.0f;
  bool done = !live;
  if (next_best < 0) done = true;
  {
    unsigned long long tm = __ballot(live && top == next_best);
    bool all_top = !done && ((unsigned)(tm >> (16 * grp)) & 0xffffu) == live_g;
    float help[NCH];
    eu_eval_facet<NCH, DEG, GEN>(p, all_top ? next_best : -1, px, tap, cx, cy, sl, 0ull, help);
    unsigned long long om = __ballot(all_top && help[NCH - 1] >= 1.0f);
    bool opaque = all_top && ((unsigned)(om >> (16 * grp)) & 0xffffu) == live_g;
    if (opaque) {
#pragma unroll
      for (int c = 0; c < NCH; c++) out[c] = help[c];
      done = true;
    }
  }
  int f = done ? -1 : top;
  float fz = tz;
  int layer = 0;
  while (true) {
    if (!__ballot(f >= 0)) break;
    float help[NCH];
    eu_eval_facet<NCH, DEG, GEN>(p, f, px, tap, cx, cy, sl, 0ull, help);
    if (f >= 0) {
      if (layer == 0) {
#pragma unroll
        for (int c = 0; c < NCH; c++) out[c] = help[c];
      } else {
        const float a = out[NCH - 1];
#pragma unroll
        for (int c = 0; c < NCH; c++) out[c] = out[c] + (1.0f - a) * help[c];
      }
      layer++;
    }
    float nz;
    const int nxt = pick(false, fz, f >= 0 ? f : 0x7fffffff, nz);     // uniform control flow: every lane runs the pass
    if (f >= 0) { f = nxt; fz = nz; }
  }
}

// _hdr_merge_syn::get_quality for a grey value (envutil_payload.cc:1388-1446); kind 0 LOW, 1 MIDDLE, 2 HIGH
__device__ __forceinline__ float eu_hdr_quality(float grey, float optimum, int kind)
{
  const bool large = grey > optimum;
  float distance = fabsf(optimum - grey);
  if (kind == 0 && !large) distance = 0.0f;
  if (kind == 2 && large) distance = 0.0f;
  const float proximity = optimum - distance;
  return proximity / (optimum * optimum);
}
__device__ __forceinline__ float eu_std_max(float a, float b) { return a < b ? b : a; }

// _hdr_merge_syn::operator() (envutil_payload.cc:1500-1622): EVERY facet is evaluated - a miss is a
// zero pixel and takes part with the quality a zero pixel has -, quality-weighted sum, normalised.
// The one per-VECTOR decision (all_of(alpha == 0) -> quality 0) is a ballot over the lane's group of 16.
template <int NCH, int DEG, bool GEN>
__device__ __forceinline__ void eu_synopsis_hdr(const eu_multi_params &p, const eu_pix &px, bool live,
                                                bool tap, float cx, float cy, float *out)
{
  constexpr bool alpha = NCH == 2 || NCH == 4;
  constexpr int ncol = alpha ? NCH - 1 : NCH;
  const int grp = (threadIdx.x & 63) >> 4;
  const unsigned live_g = (unsigned)(__ballot(live) >> (16 * grp)) & 0xffffu;
  float qsum = 0.0f;
#pragma unroll
  for (int c = 0; c < NCH; c++) out[c] = 0.0f;
#pragma unroll 1
  for (int f = 0; f < p.nfct; f++) {
    float rx, ry, rz, sx = 0.0f, sy = 0.0f;
    int face;
    eu_syn_ray<GEN>(p, f, px, tap, cx, cy, rx, ry, rz);
    const eu_src_dev &s = p.srcs[f];
    bool hit = true, any = true;
    if (!s.mask_all) {
      // whole wavefront provably outside the facet's window: its pixel is zero without the exact test
      const bool maybe = eu_multi_maybe(p, f, s, rx, ry, rz);
      hit = false;
      any = __ballot(maybe) != 0;
      if (any) hit = eu_source_coordinate(s, rx, ry, rz, sx, sy, face);
    } else {
      hit = eu_source_coordinate(s, rx, ry, rz, sx, sy, face);
    }
    float v[NCH];
#pragma unroll
    for (int c = 0; c < NCH; c++) v[c] = 0.0f;
    // a wavefront without a hit needs no gathers - unless the facet has another channel count: a miss
    // is then the ADAPTED zero pixel (repix_t gives it alpha 1)
    if (s.nch != NCH || (any && __ballot(hit))) eu_env_facet<NCH, DEG>(s, hit, sx, sy, v);
    const int kind = f == p.hdr_low ? 0 : (f == p.hdr_high ? 2 : 1);
    const float optimum = 0.5f * s.brighten;
    float grey;
    if constexpr (ncol == 1) grey = v[0];
    else grey = eu_std_max(v[0], eu_std_max(v[1], v[2]));
    float q = eu_hdr_quality(grey, optimum, kind);
    if constexpr (alpha) {
      const float a = v[NCH - 1];
      const unsigned zero_g = (unsigned)(__ballot(live && a == 0.0f) >> (16 * grp)) & 0xffffu;
      q = zero_g == live_g ? 0.0f : a * q;
    }
    qsum = qsum + q;
    if constexpr (!alpha) {
#pragma unroll
      for (int c = 0; c < NCH; c++) out[c] = out[c] + v[c] * q;
    } else {
      const float a = v[NCH - 1];
#pragma unroll
      for (int c = 0; c < ncol; c++) {
        float d = 0.0f;
        if (a > 0.000001f) d = v[c] / a;
        out[c] = out[c] + d * q;
      }
      out[NCH - 1] = eu_std_max(out[NCH - 1], a);
    }
  }
#pragma unroll
  for (int c = 0; c < ncol; c++) {
    float t = out[c] / qsum;
    if (!(qsum > 0.0f)) t = 0.0f;
    if constexpr (alpha) t = t * out[NCH - 1];
    out[c] = t;
  }
}

// The synopsis kernels are bound by the latency of their gathers (six 1-GB sources,
// little locality) more than by anything else: capping the registers for 5 waves per
// SIMD (a few spills) beats the 2-3 waves the allocator settles on by itself - config 5:
// 9.2 ms free, 8.0 at 4, 7.7 at 5, 8.8 at 6, 11.7 at 8 waves.
#ifndef EU_MULTI_WAVES
#define EU_MULTI_WAVES 5
#endif
#define EU_MULTI_OCC __attribute__((amdgpu_waves_per_eu(EU_MULTI_WAVES, EU_MULTI_WAVES)))

// GEN: some facet of the job is stepped by generic_stepper (translation, --single); only the run-time-degree
// variants are instantiated with it
// BIG: alpha compositing of more than 64 facets (eu_synopsis_big)
template <int NCH, int DEG, bool PLUS, bool HDR = false, bool GEN = false, bool BIG = false>
__global__ __launch_bounds__(256) EU_MULTI_OCC void eu_render_multi_kernel(const eu_multi_params p)
{
  extern __shared__ float eu_dyn_lds[];
  const int b = eu_xcd_tile(blockIdx.x, p.tiles_x, p.tiles_y, -EU_UNIT_ROWS);
  if (b < 0) return;
  const int tile_y = b / p.tiles_x, tile_x = b - tile_y * p.tiles_x;
  const int lane = threadIdx.x & 63;
  const int wrow = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  eu_pix px;
  px.x = tile_x * EU_TILE_W + lane;
  px.y = p.row_begin + tile_y * EU_TILE_H + wrow;
  if (px.y >= p.row_end) return;              // wave-uniform
  const bool live = px.x < p.width;
  if (!live) px.x = p.width - 1;              // keeps table reads in range; no store
  eu_slots sl;
  sl.keep = p.nfct <= EU_MULTI_KEEP;
  sl.z = eu_dyn_lds + threadIdx.x;
  sl.sx = sl.z + p.nfct * 256;
  sl.sy = sl.sx + p.nfct * 256;
  float out[NCH];
  if (!p.twine) {
    if constexpr (HDR) eu_synopsis_hdr<NCH, DEG, GEN>(p, px, live, false, 0.0f, 0.0f, out);
    else if constexpr (BIG) eu_synopsis_big<NCH, DEG, GEN>(p, px, live, false, 0.0f, 0.0f, sl, out);
    else eu_synopsis<NCH, DEG, PLUS, GEN>(p, px, live, false, 0.0f, 0.0f, sl, out);
  } else {
#pragma unroll
    for (int c = 0; c < NCH; c++) out[c] = 0.0f;
    for (int k = 0; k < p.ntaps; k++) {
      const float cx = p.taps[3 * k], cy = p.taps[3 * k + 1], cw = p.taps[3 * k + 2];
      float help[NCH];
      if constexpr (HDR) eu_synopsis_hdr<NCH, DEG, GEN>(p, px, live, true, cx, cy, help);
      else if constexpr (BIG) eu_synopsis_big<NCH, DEG, GEN>(p, px, live, true, cx, cy, sl, help);
      else eu_synopsis<NCH, DEG, PLUS, GEN>(p, px, live, true, cx, cy, sl, help);
#pragma unroll
      for (int c = 0; c < NCH; c++) out[c] = out[c] + cw * help[c];
    }
  }
  if (!live) return;
  eu_put<NCH>(p.out + (long long)(px.y - p.row_begin) * p.out_stride, px.x, out);
}

template <int NCH, bool PLUS>
static int launch_multi_n(const eu_multi_params &p, int degree, hipStream_t st)
{
  dim3 grid((unsigned)eu_xcd_grid(p.tiles_x, p.tiles_y, EU_UNIT_ROWS)), block(256);
  // alpha compositing keeps z (and, for up to EU_MULTI_KEEP facets, the source
  // coordinate) of every facet per thread in LDS
  if (PLUS && !p.hdr && p.nfct > EU_MULTI_MAXF) {
    if constexpr (PLUS) {
      eu_multi_params q = p;           // the one instantiation serves jobs with and without generic-stepper facets
      hipLaunchKernelGGL((eu_render_multi_kernel<NCH, -1, true, false, true, true>), grid, block, 0, st, q);
    }
    return hipGetLastError() == hipSuccess ? 0 : -1;
  }
  const size_t lds = PLUS && !p.hdr ? (size_t)(p.nfct <= EU_MULTI_KEEP ? 3 : 1) * p.nfct * 256 * sizeof(float) : 0;
  if (p.gen) {
    if (p.hdr) hipLaunchKernelGGL((eu_render_multi_kernel<NCH, -1, PLUS, true, true>), grid, block, lds, st, p);
    else hipLaunchKernelGGL((eu_render_multi_kernel<NCH, -1, PLUS, false, true>), grid, block, lds, st, p);
    return hipGetLastError() == hipSuccess ? 0 : -1;
  }
  if (p.hdr) {
    switch (degree) {
      case 0: hipLaunchKernelGGL((eu_render_multi_kernel<NCH, 0, PLUS, true>), grid, block, lds, st, p); break;
      case 1: hipLaunchKernelGGL((eu_render_multi_kernel<NCH, 1, PLUS, true>), grid, block, lds, st, p); break;
      case 2: hipLaunchKernelGGL((eu_render_multi_kernel<NCH, 2, PLUS, true>), grid, block, lds, st, p); break;
      case 3: hipLaunchKernelGGL((eu_render_multi_kernel<NCH, 3, PLUS, true>), grid, block, lds, st, p); break;
      default: hipLaunchKernelGGL((eu_render_multi_kernel<NCH, -1, PLUS, true>), grid, block, lds, st, p); break;
    }
    return hipGetLastError() == hipSuccess ? 0 : -1;
  }
  switch (degree) {
    case 0: hipLaunchKernelGGL((eu_render_multi_kernel<NCH, 0, PLUS>), grid, block, lds, st, p); break;
    case 1: hipLaunchKernelGGL((eu_render_multi_kernel<NCH, 1, PLUS>), grid, block, lds, st, p); break;
    case 2: hipLaunchKernelGGL((eu_render_multi_kernel<NCH, 2, PLUS>), grid, block, lds, st, p); break;
    case 3: hipLaunchKernelGGL((eu_render_multi_kernel<NCH, 3, PLUS>), grid, block, lds, st, p); break;
    default: hipLaunchKernelGGL((eu_render_multi_kernel<NCH, -1, PLUS>), grid, block, lds, st, p); break;
  }
  return hipGetLastError() == hipSuccess ? 0 : -1;
}

#endif  // EU_MULTI_NCH

#define EU_CAT2(a, b) a##b
#define EU_CAT(a, b) EU_CAT2(a, b)

#if defined(EU_MULTI_NCH) && defined(EU_MULTI_STAMPS)
extern "C" int eu_multi_stamps_read(unsigned long long *out4)
{
  static unsigned long long h[1024 * 4];
  if (hipMemcpyFromSymbol(h, HIP_SYMBOL(eu_multi_stamp_acc), sizeof h) != hipSuccess) return -1;
  for (int k = 0; k < 4; k++) { out4[k] = 0; for (int i = 0; i < 1024; i++) out4[k] += h[4 * i + k]; }
  memset(h, 0, sizeof h);
  return hipMemcpyToSymbol(HIP_SYMBOL(eu_multi_stamp_acc), h, sizeof h) == hipSuccess ? 0 : -1;
}
#endif
#ifdef EU_MULTI_NCH
// this translation unit carries the kernels of ONE channel count (the Makefile
// compiles the file four times, so that the four compile in parallel)
extern "C" int EU_CAT(eu_launch_render_multi_nch, EU_MULTI_NCH)(const eu_multi_params *p, int degree,
                                                               void *stream)
{
  constexpr bool plus = EU_MULTI_NCH == 2 || EU_MULTI_NCH == 4;
  return launch_multi_n<EU_MULTI_NCH, plus>(*p, degree, (hipStream_t)stream);
}
#else
extern "C" int eu_launch_render_multi_nch1(const eu_multi_params *p, int degree, void *stream);
extern "C" int eu_launch_render_multi_nch2(const eu_multi_params *p, int degree, void *stream);
extern "C" int eu_launch_render_multi_nch3(const eu_multi_params *p, int degree, void *stream);
extern "C" int eu_launch_render_multi_nch4(const eu_multi_params *p, int degree, void *stream);

extern "C" int eu_launch_render_multi(const void *pp, int degree, void *stream)
{
  eu_multi_params p = *(const eu_multi_params *)pp;
  // voronoi_syn and hdr_merge keep no per-facet state; alpha compositing beyond 64 facets: eu_synopsis_big
  p.tiles_x = (p.width + EU_TILE_W - 1) / EU_TILE_W;
  p.tiles_y = (p.row_end - p.row_begin + EU_TILE_H - 1) / EU_TILE_H;
  if (p.tiles_x <= 0 || p.tiles_y <= 0) return 0;
  switch (p.nch) {
    case 1: return eu_launch_render_multi_nch1(&p, degree, stream);
    case 2: return eu_launch_render_multi_nch2(&p, degree, stream);
    case 3: return eu_launch_render_multi_nch3(&p, degree, stream);
    case 4: return eu_launch_render_multi_nch4(&p, degree, stream);
  }
  return -2;
}
#endif

// Set-up stages on the device: what envutil does on the CPU before the render
// loop can run - b-spline prefilter + bracing of a source image
// (zimt/prefilter.h, recursive.h, brace.h; environment.h:356-522) and the
// cubemap "IR" image with its support frame (cubemap.h:576-946).
//
// Every 1-D line is filtered by one thread with exactly the float operations of
// zimt's iir_filter::solve_gain_inlined (recursive.h:631-733); lines are
// independent, so the grid is "one thread per (line, channel)".
// Compiled with -ffp-contract=off.

#include <hip/hip_runtime.h>
#include <cfloat>
#include <climits>
#include <cmath>
#include "eu_device.h"
#include "eu_setup_math.h"
#include "eu_math.h"

namespace {

#define EU_MAX_POLES (EU_MAX_DEGREE / 2 + 1)

struct iir_dev {
  int bc, npoles;
  float pole[EU_MAX_POLES];
  float zpow[EU_MAX_POLES];   // closed-form branches: pole^(M-1) or pole^(2M), in float
  int horizon[EU_MAX_POLES];
  float gain;
};

// recursive.h:790-830 (horizon, gain) - long double on the host
iir_dev make_iir(int bc, int degree, long double tolerance, long M)
{
  iir_dev f;
  memset(&f, 0, sizeof f);
  long double lp[EU_MAX_POLES];
  f.bc = bc;
  f.npoles = eu::poles(degree, lp);
  long double gain = 1.0L;
  for (int k = 0; k < f.npoles; k++) {
    f.pole[k] = (float)lp[k];
    f.horizon[k] = tolerance > 0
      ? (int)ceill(logl(tolerance) / logl(fabsl(lp[k]))) : INT_MAX;
    gain *= (1.0L - lp[k]) * (1.0L - 1.0L / lp[k]);
    // icc_natural/icc_mirror use pole^(M-1), icc_reflect pole^(2M), taken in
    // long double and narrowed (recursive.h:335, :413, :483)
    long double e = bc == EU_BC_REFLECT ? (long double)(2 * M) : (long double)(M - 1);
    f.zpow[k] = (float)powl(lp[k], e);
  }
  f.gain = (float)gain;
  return f;
}

// a line: element n lives at base[off(n)]
struct strided_line {
  float *base; long long es;
  __device__ float get(int n) const { return base[(long long)n * es]; }
  __device__ void put(int n, float v) const { base[(long long)n * es] = v; }
};

// environment.h:395-447: column x of the left half top->bottom, then column
// x + W/2 bottom->top, as one line of length 2H
struct stacked_line {
  float *up, *down; long long es; int H;
  __device__ long long off(int n) const { return n < H ? (long long)n * es : 0; }
  __device__ float get(int n) const { return n < H ? up[(long long)n * es] : down[(long long)(2 * H - 1 - n) * es]; }
  __device__ void put(int n, float v) const { if (n < H) up[(long long)n * es] = v; else down[(long long)(2 * H - 1 - n) * es] = v; }
};

// recursive.h:321-620
template <class L>
__device__ float icc(const iir_dev &f, const L &c, int M, int k)
{
  float z = f.pole[k], zn, z2n, iz, Sum;
  int n, hz = f.horizon[k];
  switch (f.bc) {
    case EU_BC_NATURAL:
      if (hz < M) {
        float c02 = c.get(0) + c.get(0);
        zn = z; Sum = c.get(0);
        for (n = 1; n < hz; n++) { Sum = Sum + zn * (c02 - c.get(n)); zn = zn * z; }
        return Sum;
      }
      zn = z; iz = 1.0f / z; z2n = f.zpow[k];
      Sum = ((1.0f + z) / (1.0f - z)) * (c.get(0) - z2n * c.get(M - 1));
      z2n = z2n * (z2n * iz);
      for (n = 1; n <= M - 2; n++) { Sum = Sum - (zn - z2n) * c.get(n); zn = zn * z; z2n = z2n * iz; }
      return Sum / (1.0f - zn * zn);
    case EU_BC_REFLECT:
      if (hz < M) {
        zn = z; Sum = c.get(0);
        for (n = 0; n < hz; n++) { Sum = Sum + zn * c.get(n); zn = zn * z; }
        return Sum;
      }
      zn = z; iz = 1.0f / z; z2n = f.zpow[k];
      Sum = 0.0f;
      for (n = 0; n < M - 1; n++) { Sum = Sum + (zn + z2n) * c.get(n); zn = zn * z; z2n = z2n * iz; }
      Sum = Sum + (zn + z2n) * c.get(n);
      return c.get(0) + Sum / (1.0f - zn * zn);
    case EU_BC_PERIODIC:
      if (hz < M) {
        zn = z; Sum = c.get(0);
        for (n = M - 1; n > (M - hz); n--) { Sum = Sum + zn * c.get(n); zn = zn * z; }
      } else {
        zn = z; Sum = c.get(0);
        for (n = M - 1; n > 0; n--) { Sum = Sum + zn * c.get(n); zn = zn * z; }
        Sum = Sum / (1.0f - zn);
      }
      return Sum;
    case EU_BC_MIRROR:
      if (hz < M) {
        zn = z; Sum = c.get(0);
        for (n = 1; n < hz; n++) { Sum = Sum + zn * c.get(n); zn = zn * z; }
        return Sum;
      }
      zn = z; iz = 1.0f / z; z2n = f.zpow[k];
      Sum = c.get(0) + z2n * c.get(M - 1);
      z2n = z2n * (z2n * iz);
      for (n = 1; n <= M - 2; n++) { Sum = Sum + (zn + z2n) * c.get(n); zn = zn * z; z2n = z2n * iz; }
      return Sum / (1.0f - zn * zn);
    default:
      return c.get(0);
  }
}

template <class L>
__device__ float iacc(const iir_dev &f, const L &c, int M, int k)
{
  float z = f.pole[k], zn, Sum;
  int hz = f.horizon[k];
  switch (f.bc) {
    case EU_BC_NATURAL:
      return -(z / ((1.0f - z) * (1.0f - z))) * (c.get(M - 1) - z * c.get(M - 2));
    case EU_BC_REFLECT:
      return c.get(M - 1) / (1.0f - 1.0f / z);
    case EU_BC_PERIODIC:
      if (hz < M) {
        zn = z; Sum = c.get(M - 1) * z;
        for (int n = 0; n < hz; n++) { zn = zn * z; Sum = Sum + zn * c.get(n); }
        Sum = -Sum;
      } else {
        zn = z; Sum = c.get(M - 1);
        for (int n = 0; n < M - 1; n++) { Sum = Sum + zn * c.get(n); zn = zn * z; }
        Sum = z * Sum / (zn - 1.0f);
      }
      return Sum;
    case EU_BC_MIRROR:
      return (z / (z * z - 1.0f)) * (c.get(M - 1) + z * c.get(M - 2));
    default:
      return c.get(M - 1);
  }
}

// The recursions are sequential by definition (and stay so: same operations in the same order
// as recursive.h), but their LOADS are not. A launch has one thread per line - 24 576 lines for
// the 16K x 8K RGB source, 1.5 wavefronts per CU - so what a sweep achieves is (bytes a thread
// keeps in flight) / (memory latency): a line is walked in blocks of EU_IIR_BLOCK samples, and
// the loads of block k + 1 are issued before the dependent chain of block k runs (two register
// buffers). 16 samples, one buffer: 0.9 TB/s (4.1 + 3.0 ms for the two passes of that source);
// profiles/r02_prefilter.txt has the numbers for this form.
#ifndef EU_IIR_BLOCK
#define EU_IIR_BLOCK 64
#endif

template <class L>
__device__ __forceinline__ void iir_load(const L &x, int n, int dir, float *v)
{
#pragma unroll
  for (int i = 0; i < EU_IIR_BLOCK; i++) v[i] = x.get(n + dir * i);
}
template <class L>
__device__ __forceinline__ void iir_store(const L &x, int n, int dir, const float *v)
{
#pragma unroll
  for (int i = 0; i < EU_IIR_BLOCK; i++) x.put(n + dir * i, v[i]);
}

// forward: X = gain * x[n] + p * X (gain == 1 and the product skipped for k > 0)
template <bool GAIN, class L>
__device__ __forceinline__ float causal_pass(const L &x, int M, float g, float p, float X)
{
  int n = 1;
  if (n + EU_IIR_BLOCK <= M) {
    float a[EU_IIR_BLOCK], b[EU_IIR_BLOCK];
    iir_load(x, n, 1, a);
    for (;;) {
      // block at n is in a; request the one behind it into b, then run a's chain
      const bool more = n + 2 * EU_IIR_BLOCK <= M;
      if (more) iir_load(x, n + EU_IIR_BLOCK, 1, b);
#pragma unroll
      for (int i = 0; i < EU_IIR_BLOCK; i++) {
        if constexpr (GAIN) X = g * a[i] + p * X;
        else X = a[i] + p * X;
        a[i] = X;
      }
      iir_store(x, n, 1, a);
      n += EU_IIR_BLOCK;
      if (!more) break;
      const bool more2 = n + 2 * EU_IIR_BLOCK <= M;
      if (more2) iir_load(x, n + EU_IIR_BLOCK, 1, a);
#pragma unroll
      for (int i = 0; i < EU_IIR_BLOCK; i++) {
        if constexpr (GAIN) X = g * b[i] + p * X;
        else X = b[i] + p * X;
        b[i] = X;
      }
      iir_store(x, n, 1, b);
      n += EU_IIR_BLOCK;
      if (!more2) break;
    }
  }
  for (; n < M; n++) {
    if constexpr (GAIN) X = g * x.get(n) + p * X;
    else X = x.get(n) + p * X;
    x.put(n, X);
  }
  return X;
}

// backward from M - 2: X = p * (X - x[n])
template <class L>
__device__ __forceinline__ void anticausal_pass(const L &x, int M, float p, float X)
{
  int n = M - 2;
  if (n - (EU_IIR_BLOCK - 1) >= 0) {
    float a[EU_IIR_BLOCK], b[EU_IIR_BLOCK];
    iir_load(x, n, -1, a);
    for (;;) {
      const bool more = n - (2 * EU_IIR_BLOCK - 1) >= 0;
      if (more) iir_load(x, n - EU_IIR_BLOCK, -1, b);
#pragma unroll
      for (int i = 0; i < EU_IIR_BLOCK; i++) { X = p * (X - a[i]); a[i] = X; }
      iir_store(x, n, -1, a);
      n -= EU_IIR_BLOCK;
      if (!more) break;
      const bool more2 = n - (2 * EU_IIR_BLOCK - 1) >= 0;
      if (more2) iir_load(x, n - EU_IIR_BLOCK, -1, a);
#pragma unroll
      for (int i = 0; i < EU_IIR_BLOCK; i++) { X = p * (X - b[i]); b[i] = X; }
      iir_store(x, n, -1, b);
      n -= EU_IIR_BLOCK;
      if (!more2) break;
    }
  }
  for (; n >= 0; n--) { X = p * (X - x.get(n)); x.put(n, X); }
}

// recursive.h:631-733, in place
template <class L>
__device__ void solve_line(const iir_dev &f, const L &x, int M)
{
  if (M == 1 || f.npoles < 1) return;
  float p = f.pole[0], g = f.gain;
  float X = g * icc(f, x, M, 0);
  x.put(0, X);
  causal_pass<true>(x, M, g, p, X);
  X = iacc(f, x, M, 0);
  x.put(M - 1, X);
  anticausal_pass(x, M, p, X);
  for (int k = 1; k < f.npoles; k++) {
    p = f.pole[k];
    X = icc(f, x, M, k);
    x.put(0, X);
    causal_pass<false>(x, M, 1.0f, p, X);
    X = iacc(f, x, M, k);
    x.put(M - 1, X);
    anticausal_pass(x, M, p, X);
  }
}

// lines: nl x nch threads; line i channel c starts at base + i*line_stride + c
__global__ __launch_bounds__(64) void filter_lines_kernel(iir_dev f, float *base, long long nl, int nch,
                                    long long line_stride, int len, long long es)
{
  long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= nl * nch) return;
  long long i = t / nch;
  int c = (int)(t - i * nch);
  strided_line ln { base + i * line_stride + c, es };
  solve_line(f, ln, len);
}

__global__ __launch_bounds__(64) void filter_stacked_kernel(iir_dev f, float *core, long long count,
                                      long long down_off, long long row_es, int H)
{
  long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= count) return;   // t = x * nch + c
  stacked_line ln { core + t, core + t + down_off, row_es, H };
  solve_line(f, ln, 2 * H);
}

// ---------------------------------------------------------------------------
// The same sweeps, streamed. The recursions above move every sample with one 4-byte access per
// lane; in the row direction the 64 lanes of a wavefront then touch 64 different cache lines
// per instruction (64 cycles of the texture addresser each: 16 384 samples x 4 accesses x 64
// cycles = 1.7 ms per wavefront is what the row pass of the 16K x 8K source cost), and in the
// column direction a wavefront has at most 63 such accesses in flight. Here ONE wavefront per
// workgroup walks its lines in blocks of 64 samples through LDS tiles:
//   * LDS-DMA (global_load_lds_dwordx4: 16 bytes per lane straight into LDS, no registers)
//     requests block b + EU_IIR_AHEAD while block b is worked on,
//   * the lanes that own a line run the recursion on the tile in place (ds_read / ds_write,
//     the operations and their order are those of causal_pass / anticausal_pass),
//   * the tile goes back with 16-byte stores.
// What remains serial is the recursion itself: 3 VALU + 2 LDS instructions per sample from a
// single wavefront. Lanes per wavefront are kept LOW (8 rows x nch, 32 columns) so that the
// 24 576 lines of that source make ~1000 wavefronts, one per SIMD of the chip.
// vmcnt counts LDS-DMA and stores in issue order; every vector memory instruction inside the
// block loops is inline asm, so the counts below are exact.
// ---------------------------------------------------------------------------
#ifndef EU_IIR_AHEAD
#define EU_IIR_AHEAD 3
#endif
#ifdef EU_IIR_EXPERIMENT   // timing experiments only (results are wrong): 1 no recursion, 2 no requests, 4 no write-back
__device__ int iir_exp;
#define IIR_EXP(bit) (iir_exp & (bit))
#else
#define IIR_EXP(bit) 0
#endif
#define EU_IIR_BUFS (EU_IIR_AHEAD + 1)

typedef __attribute__((address_space(3))) float *iir_lptr;
typedef __attribute__((address_space(3))) void *iir_lvoid;
typedef float iir_f4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) iir_f4 *iir_l4ptr;

__device__ __forceinline__ void iir_dma16(unsigned dst, const float *src)
{
  unsigned keep;
  asm volatile("s_mov_b32 %0, m0\n\t"
               "s_mov_b32 m0, %1\n\t"
               "s_nop 0\n\t"
               "global_load_lds_dwordx4 %2, off\n\t"
               "s_mov_b32 m0, %0"
               : "=&s"(keep) : "s"(dst), "v"(src) : "memory");
}
__device__ __forceinline__ void iir_store16(float *dst, iir_f4 v)
{
  // the two wait states a VALU write of the data registers must keep from a store of more than
  // 8 bytes (the compiler's hazard recogniser does not look into inline asm)
  asm volatile("global_store_dwordx4 %0, %1, off\n\ts_nop 1" :: "v"(dst), "v"(v) : "memory");
}
template <int N> __device__ __forceinline__ void iir_wait_vm()
{
  static_assert(N >= 0 && N <= 63, "vmcnt holds 6 bits");
  asm volatile("s_waitcnt vmcnt(%0)" :: "n"(N) : "memory");
}

// R rows x NCH channels per wavefront; a block is 64 samples = 64*NCH floats of every row.
// LDS pitch 64*NCH + 4 floats: the chain lanes (row i, channel c) sit at bank 4i + c + NCH*n.
template <int NCH, int R> struct tile_rows {
  static constexpr int STRIDE = NCH, PITCH = 64 * NCH + 4, FLOATS = R * PITCH, OPS = R;
  static_assert(R <= 16 && R * NCH <= 64, "bank pattern / lanes");
  float *g;          // this lane's 16-byte chunk of row 0, block 0
  long long ls;      // floats between rows
  int lane;
  __device__ bool mover() const { return lane < 16 * NCH; }
  __device__ int chain_off() const { return (lane / NCH) * PITCH + lane % NCH; }
  __device__ void request(unsigned lds, int blk) const
  {
    if (mover() && !IIR_EXP(2)) {
      const float *s = g + (long long)blk * (64 * NCH);
#pragma unroll
      for (int r = 0; r < R; r++) iir_dma16(lds + r * (PITCH * 4), s + r * ls);
    }
  }
  __device__ void writeback(iir_lptr t, int blk) const
  {
    if (mover() && !IIR_EXP(4)) {
      float *d = g + (long long)blk * (64 * NCH);
      iir_f4 v[R];
#pragma unroll
      for (int r = 0; r < R; r++) v[r] = *(iir_l4ptr)(t + r * PITCH + lane * 4);
#pragma unroll
      for (int r = 0; r < R; r++) iir_store16(d + r * ls, v[r]);
    }
  }
};

// L adjacent floats of every row per wavefront (L/4 chunks x 256/L rows per instruction); with
// STACKED sample n >= H is row 2H-1-n of the other half (stacked_line)
template <int L, bool STACKED> struct tile_cols {
  static constexpr int STRIDE = L, FLOATS = 64 * L, OPS = L / 4, RPI = 256 / L;
  float *up, *down;  // this lane's chunk in row 0
  long long es; int H;
  int lane;
  __device__ int chain_off() const { return lane; }
  __device__ float *rowptr(int n) const
  {
    if (STACKED && n >= H) return down + (long long)(2 * H - 1 - n) * es;
    return up + (long long)n * es;
  }
  __device__ void request(unsigned lds, int blk) const
  {
    const int n0 = blk * 64 + lane / (L / 4);
    if (IIR_EXP(2)) return;
#pragma unroll
    for (int q = 0; q < OPS; q++) iir_dma16(lds + q * 1024, rowptr(n0 + q * RPI));
  }
  __device__ void writeback(iir_lptr t, int blk) const
  {
    const int n0 = blk * 64 + lane / (L / 4);
    if (IIR_EXP(4)) return;
    iir_f4 v[OPS];
#pragma unroll
    for (int q = 0; q < OPS; q++) v[q] = *(iir_l4ptr)(t + q * 256 + lane * 4);
#pragma unroll
    for (int q = 0; q < OPS; q++) iir_store16(rowptr(n0 + q * RPI), v[q]);
  }
};

// one LDS buffer: the tile and, behind it, the 64 checkpoints of the block (one per lane)
template <class T> struct iir_bufs {
  static constexpr int BUF = T::FLOATS + 64;
  static __device__ __forceinline__ unsigned at(unsigned lds0, int k) { return lds0 + (unsigned)(k % EU_IIR_BUFS) * (unsigned)(BUF * 4); }
};

__device__ __forceinline__ void iir_dma4(unsigned dst, const float *src)
{
  unsigned keep;
  asm volatile("s_mov_b32 %0, m0\n\t"
               "s_mov_b32 m0, %1\n\t"
               "s_nop 0\n\t"
               "global_load_lds_dword %2, off\n\t"
               "s_mov_b32 m0, %0"
               : "=&s"(keep) : "s"(dst), "v"(src) : "memory");
}
__device__ __forceinline__ void iir_store4(float *dst, float v)
{
  asm volatile("global_store_dword %0, %1, off" :: "v"(dst), "v"(v) : "memory");
}

// Forward sweep over blocks 0 .. nblk-1; sample 0 becomes X (the initial coefficient).
// ckpt == nullptr: every block is written back (the anticausal sweep then reads the causal
// result). Otherwise only the first and the last block are - the anticausal initial value
// (iacc) reads from them - and of every block the value the recursion ENTERS it with is kept:
// ckpt[64 * b + lane]. The backward sweep recomputes the causal result of a block from the
// input and that value, with the same operations in the same order: one pass over the image
// less (3 instead of 4 per axis) for one more recursion per sample.
// vmcnt: the waits name a count that is never larger than the number of operations issued
// behind the block's requests (extra stores only make a wait end a little later).
template <bool GAIN, class T>
__device__ __forceinline__ float stream_causal(const T &tl, float *smem, float *ckpt, int nblk, bool active,
                                               float g, float p, float X)
{
  typedef iir_bufs<T> B;
  const unsigned lds0 = (unsigned)(unsigned long long)(iir_lvoid)smem;
  for (int b = 0; b < EU_IIR_AHEAD && b < nblk; b++) tl.request(B::at(lds0, b), b);
  for (int b = 0; b < nblk; b++) {
    if (b + EU_IIR_AHEAD < nblk) {
      tl.request(B::at(lds0, b + EU_IIR_AHEAD), b + EU_IIR_AHEAD);
      if (!ckpt && b >= EU_IIR_AHEAD) iir_wait_vm<2 * T::OPS * EU_IIR_AHEAD>();
      else iir_wait_vm<T::OPS * EU_IIR_AHEAD>();
    } else {
      iir_wait_vm<0>();
    }
    const bool keep = !ckpt || b == 0 || b == nblk - 1;
    iir_lptr tile = (iir_lptr)smem + (b % EU_IIR_BUFS) * B::BUF;
    if (ckpt && active) iir_store4(ckpt + b * 64, X);
    if (active && !IIR_EXP(1)) {
      iir_lptr t = tile + tl.chain_off();
      float a[64];
#pragma unroll
      for (int i = 0; i < 64; i++) a[i] = t[i * T::STRIDE];
      {
        float Y;
        if constexpr (GAIN) Y = g * a[0] + p * X; else Y = a[0] + p * X;
        X = b == 0 ? X : Y;
        a[0] = X;
      }
#pragma unroll
      for (int i = 1; i < 64; i++) {
        if constexpr (GAIN) X = g * a[i] + p * X; else X = a[i] + p * X;
        a[i] = X;
      }
      if (keep) {
#pragma unroll
        for (int i = 0; i < 64; i++) t[i * T::STRIDE] = a[i];
      }
    }
    if (keep) tl.writeback(tile, b);
  }
  return X;
}

// Backward sweep over blocks nblk-1 .. 0; with `first` sample 64*nblk-1 becomes X (it is the
// line's last). With ckpt the blocks between the first and the last hold the INPUT: their
// causal result is recomputed from the checkpoint before the anticausal recursion runs over it.
template <bool GAIN, class T>
__device__ __forceinline__ void stream_anticausal(const T &tl, float *smem, const float *ckpt, int nblk, bool active,
                                                  float g, float p, float X, bool first)
{
  typedef iir_bufs<T> B;
  const unsigned lds0 = (unsigned)(unsigned long long)(iir_lvoid)smem;
  auto request = [&](int k, int b) {
    tl.request(B::at(lds0, k), b);
    if (ckpt) iir_dma4(B::at(lds0, k) + T::FLOATS * 4, ckpt + b * 64);
  };
  for (int j = 0; j < EU_IIR_AHEAD && j < nblk; j++) request(j, nblk - 1 - j);
  for (int j = 0; j < nblk; j++) {
    const int b = nblk - 1 - j;
    if (j + EU_IIR_AHEAD < nblk) {
      request(j + EU_IIR_AHEAD, b - EU_IIR_AHEAD);
      if (j >= EU_IIR_AHEAD) iir_wait_vm<2 * T::OPS * EU_IIR_AHEAD>();
      else iir_wait_vm<T::OPS * EU_IIR_AHEAD>();
    } else {
      iir_wait_vm<0>();
    }
    const bool redo = ckpt && b != 0 && b != nblk - 1;
    iir_lptr tile = (iir_lptr)smem + (j % EU_IIR_BUFS) * B::BUF;
    if (active && !IIR_EXP(1)) {
      iir_lptr t = tile + tl.chain_off();
      float a[64];
#pragma unroll
      for (int i = 63; i >= 0; i--) a[i] = t[i * T::STRIDE];
      if (redo) {
        float C = tile[T::FLOATS + tl.lane];
#pragma unroll
        for (int i = 0; i < 64; i++) {
          if constexpr (GAIN) C = g * a[i] + p * C; else C = a[i] + p * C;
          a[i] = C;
        }
      }
      {
        float Y = p * (X - a[63]);
        X = (first && j == 0) ? X : Y;
        a[63] = X;
      }
#pragma unroll
      for (int i = 62; i >= 0; i--) { X = p * (X - a[i]); a[i] = X; }
#pragma unroll
      for (int i = 63; i >= 0; i--) t[i * T::STRIDE] = a[i];
    }
    tl.writeback(tile, b);
  }
}

// solve_line with the block part of every sweep streamed; M >= 64. ckpt: this lane's
// checkpoint column (64 floats per block) or nullptr
template <class Ln, class T>
__device__ void solve_line_stream(const iir_dev &f, const Ln &x, const T &tl, int M, bool active, float *smem,
                                  float *ckpt)
{
  const int nblk = M / 64, n1 = nblk * 64;
  const float g = f.gain;
  for (int k = 0; k < f.npoles; k++) {
    const float p = f.pole[k];
    float X = 0.0f;
    if (active) { X = icc(f, x, M, k); if (k == 0) X = g * X; }
    if (k == 0) X = stream_causal<true>(tl, smem, ckpt, nblk, active, g, p, X);
    else X = stream_causal<false>(tl, smem, ckpt, nblk, active, 1.0f, p, X);
    if (active)
      for (int n = n1; n < M; n++) {
        if (k == 0) X = g * x.get(n) + p * X; else X = x.get(n) + p * X;
        x.put(n, X);
      }
    iir_wait_vm<0>();
    const bool tail = n1 < M;
    if (active) {
      X = iacc(f, x, M, k);
      if (tail) {
        x.put(M - 1, X);
        for (int n = M - 2; n >= n1; n--) { X = p * (X - x.get(n)); x.put(n, X); }
      }
    }
    if (k == 0) stream_anticausal<true>(tl, smem, ckpt, nblk, active, g, p, X, !tail);
    else stream_anticausal<false>(tl, smem, ckpt, nblk, active, 1.0f, p, X, !tail);
    iir_wait_vm<0>();
  }
}

// Workgroup b of a launch runs on XCD b % 8 (round-robin dispatch), and every XCD has its own
// L2. Column groups that are neighbours in memory share the cache lines at their common edge
// (a row of the container does not start on a line boundary): give every XCD a CONTIGUOUS range
// of groups, so that both halves of such a line are read and - what matters - written through
// the same L2 within a few microseconds and leave it as one full line.
__device__ __forceinline__ unsigned iir_xcd_group(unsigned b, unsigned n)
{
  const unsigned q = n / 8, r = n % 8, xcd = b % 8, idx = b / 8;
  return xcd * q + (xcd < r ? xcd : r) + idx;
}

// A workgroup is blockDim.x / 64 INDEPENDENT wavefronts (no barrier anywhere), each with its own
// group of lines and its own slice of the dynamic LDS.
// ckpt: groups x (len / 64) x 64 floats of scratch, or nullptr
extern __shared__ __attribute__((aligned(16))) float iir_dyn_lds[];

template <int NCH, int R>
__global__ __launch_bounds__(256) void filter_rows_stream_kernel(iir_dev f, float *base, long long line_stride, int len,
                                                                float *ckpt, unsigned ngroups)
{
  const int lane = threadIdx.x % 64, wave = __builtin_amdgcn_readfirstlane(threadIdx.x / 64), wpg = blockDim.x / 64;
  const unsigned group = blockIdx.x * wpg + wave;
  if (group >= ngroups) return;
  float *smem = iir_dyn_lds + wave * (EU_IIR_BUFS * iir_bufs<tile_rows<NCH, R>>::BUF);
  float *rows = base + (long long)group * R * line_stride;
  tile_rows<NCH, R> tl { rows + lane * 4, line_stride, lane };
  const bool active = lane < R * NCH;
  const int li = active ? lane / NCH : 0, c = lane % NCH;
  strided_line ln { rows + li * line_stride + c, NCH };
  if (ckpt) ckpt += (long long)group * (len / 64) * 64 + lane;
  solve_line_stream(f, ln, tl, len, active, smem, ckpt);
}

template <int L>
__global__ __launch_bounds__(256) void filter_cols_stream_kernel(iir_dev f, float *base, long long es, int len, float *ckpt,
                                                                unsigned ngroups)
{
  const int lane = threadIdx.x % 64, wave = __builtin_amdgcn_readfirstlane(threadIdx.x / 64), wpg = blockDim.x / 64;
  const unsigned group = iir_xcd_group(blockIdx.x, gridDim.x) * wpg + wave;
  if (group >= ngroups) return;
  float *smem = iir_dyn_lds + wave * (EU_IIR_BUFS * iir_bufs<tile_cols<L, false>>::BUF);
  float *cols = base + (long long)group * L;
  tile_cols<L, false> tl { cols + (lane % (L / 4)) * 4, nullptr, es, 0, lane };
  const bool active = lane < L;
  strided_line ln { cols + (active ? lane : 0), es };
  if (ckpt) ckpt += (long long)group * (len / 64) * 64 + lane;
  solve_line_stream(f, ln, tl, len, active, smem, ckpt);
}

template <int L>
__global__ __launch_bounds__(256) void filter_stacked_stream_kernel(iir_dev f, float *core, long long down_off,
                                                                   long long row_es, int H, float *ckpt, unsigned ngroups)
{
  const int lane = threadIdx.x % 64, wave = __builtin_amdgcn_readfirstlane(threadIdx.x / 64), wpg = blockDim.x / 64;
  const unsigned group = iir_xcd_group(blockIdx.x, gridDim.x) * wpg + wave;
  if (group >= ngroups) return;
  float *smem = iir_dyn_lds + wave * (EU_IIR_BUFS * iir_bufs<tile_cols<L, true>>::BUF);
  float *cols = core + (long long)group * L;
  float *mine = cols + (lane % (L / 4)) * 4;
  tile_cols<L, true> tl { mine, mine + down_off, row_es, H, lane };
  const bool active = lane < L;
  float *own = cols + (active ? lane : 0);
  stacked_line ln { own, own + down_off, row_es, H };
  if (ckpt) ckpt += (long long)group * (2 * H / 64) * 64 + lane;
  solve_line_stream(f, ln, tl, 2 * H, active, smem, ckpt);
}

// zimt/brace.h:134-330 for one axis; slices span the whole container
__device__ __forceinline__ long long brace_source(int bc, long long lsz, long long m,
                                                  long long i, bool left, bool *natural,
                                                  long long *pivot)
{
  long long l0 = lsz - 1, r0 = lsz + m;
  *natural = false;
  *pivot = left ? l0 + 1 : r0 - 1;
  if (m == 1) return lsz;
  switch (bc) {
    case EU_BC_PERIODIC: return left ? l0 + m - i : r0 - m + i;
    case EU_BC_NATURAL:  *natural = true;  /* fall through */
    case EU_BC_MIRROR:   return left ? l0 + 2 + i : r0 - 2 - i;
    case EU_BC_REFLECT:  return left ? l0 + 1 + i : r0 - 1 - i;
    default:             return *pivot;          // CONSTANT
  }
}

// axis 0: one thread per (frame column slot, container row, channel)
__global__ void brace_kernel(float *data, long long sx, long long sy, int nch, int axis,
                             int bc, long long lsz, long long rsz)
{
  const long long len = axis == 0 ? sx : sy, other = axis == 0 ? sy : sx;
  const long long m = len - lsz - rsz;
  const long long nframe = lsz + rsz;
  long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= nframe * other * nch) return;
  int c = (int)(t % nch);
  long long r = t / nch;
  long long o, fslot;
  if (axis == 0) { fslot = r % nframe; o = r / nframe; }
  else           { o = r % other; fslot = r / other; }
  bool left = fslot < lsz;
  long long i = left ? fslot : fslot - lsz;
  long long target = left ? (lsz - 1 - i) : (lsz + m + i);
  bool natural; long long pivot;
  long long src = brace_source(bc, lsz, m, i, left, &natural, &pivot);
  auto at = [&](long long a) -> float * {
    return axis == 0 ? data + (o * sx + a) * nch + c : data + (a * sx + o) * nch + c;
  };
  if (bc == EU_BC_ZEROPAD && m != 1) { *at(target) = 0.0f; return; }
  if (natural && m != 1) { float a = *at(pivot), b = *at(src); *at(target) = a + a - b; }
  else *at(target) = *at(src);
}

// environment.h:452-516: frame rows above/below a full spherical image come
// from the opposite meridian
__global__ void pole_rows_kernel(float *core, long long W, long long H, long long pitch,
                                 int nch, long long top, long long bottom)
{
  long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  long long n = (top + bottom) * W * nch;
  if (t >= n) return;
  int c = (int)(t % nch);
  long long r = t / nch, x = r % W, k = r / W;
  long long half = W / 2;
  long long xs = x < half ? x + half : x - half;
  long long yt, ys;
  if (k < top) { yt = -1 - k; ys = k; }
  else { long long kk = k - top; yt = H + kk; ys = H - 1 - kk; }
  if (W & 1) return;
  core[(yt * pitch + x) * nch + c] = core[(ys * pitch + xs) * nch + c];
}

// the same for an image LOWER than its frame: the reference's loop (environment.h:455-516) fills one row above and
// one row below per round, each from a source row that only has to lie inside the container - the sources run on
// into frame rows written in earlier rounds. One thread per column pair (x, x + W/2) and channel walks the rounds.
__global__ void pole_rows_seq_kernel(float *core, long long W, long long H, long long pitch, int nch,
                                     long long top, long long bottom)
{
  const long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  const long long half = W / 2;
  if ((W & 1) || t >= half * nch) return;
  const int c = (int)(t % nch);
  const long long x = t / nch;
  const long long y0 = -top, y1 = H + bottom;
  long long us = 0, ut = -1, ls = H - 1, lt = H;
  auto at = [&](long long xx, long long yy) -> float & { return core[(yy * pitch + xx) * nch + c]; };
  while (true) {
    const bool c1 = us >= y0 && us < y1, c2 = ut >= y0 && ut < y1, c3 = ls >= y0 && ls < y1, c4 = lt >= y0 && lt < y1;
    if (!c2 && !c4) break;
    if (c2) {
      if (c1) { const float a = at(x + half, us), b = at(x, us); at(x, ut) = a; at(x + half, ut) = b; us++; }
      ut--;
    }
    if (c4) {
      if (c3) { const float a = at(x + half, ls), b = at(x, ls); at(x, lt) = a; at(x + half, lt) = b; ls--; }
      lt++;
    }
  }
}

// ---- cubemap IR ------------------------------------------------------------

__global__ void place_faces_kernel(const float *faces, float *ir, int nch, long long F,
                                   long long S, long long lf)
{
  long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= 6 * F * F * nch) return;
  int c = (int)(t % nch);
  long long r = t / nch, x = r % F, yy = r / F, f = yy / F, y = yy % F;
  ir[((f * S + lf + y) * S + lf + x) * nch + c] = faces[t];
}

// cubemap.h:607-659. The four corner pixels of the ring are written twice in
// the reference (x loop, then y loop); the second write wins: corner (-1,-1)
// ends up as a copy of (0,-1), which itself is a copy of (0,0).
__global__ void mirror_around_kernel(float *ir, int nch, long long F, long long S,
                                     long long lf, long long rf)
{
  long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  const long long per = F + 2;            // positions -1 .. F
  if (t >= 6 * 4 * per * nch) return;
  int c = (int)(t % nch);
  long long r = t / nch, pos = r % per - 1, side = (r / per) % 4, f = r / (4 * per);
  int cmin = lf > 0 ? -1 : 0, cmax = rf > 0 ? (int)F : (int)F - 1;
  if (pos < cmin || pos > cmax) return;
  float *cf = ir + ((f * S + lf) * S + lf) * nch + c;
  auto px = [&](long long x, long long y) -> float * { return cf + (y * S + x) * nch; };
  // sides 0/1: top/bottom rows (x loop); 2/3: left/right columns (y loop).
  // Corner positions of sides 0/1 are overwritten by sides 2/3, so skip them
  // there and let the column pass source them from the row the x loop filled.
  bool corner = pos == -1 || pos == F;
  switch (side) {
    case 0: if (lf && !corner) *px(pos, -1) = *px(pos, 0); break;
    case 1: if (rf && !corner) *px(pos, F) = *px(pos, F - 1); break;
    case 2:
      if (lf) {
        long long sy = pos == -1 ? 0 : (pos == F ? F - 1 : pos);
        *px(-1, pos) = *px(0, sy);
      }
      break;
    default:
      if (rf) {
        long long sy = pos == -1 ? 0 : (pos == F ? F - 1 : pos);
        *px(F, pos) = *px(F - 1, sy);
      }
  }
}

// The same bracing in zimt's own ORDER (brace.h:134-330): slice after slice outward, left and right
// alternating, each slice from the slice the rule names AS IT STANDS - for a core narrower than the
// frame the rule runs into slices filled a step before, which the all-at-once kernel above cannot
// reproduce. One thread per line along the axis; used for such cores only (a few pixels wide).
__global__ void brace_seq_kernel(float *data, long long sx, long long sy, int nch, int axis, int bc,
                                 long long lsz, long long rsz)
{
  const long long w = axis == 0 ? sx : sy, other = axis == 0 ? sy : sx;
  const long long m = w - lsz - rsz;
  long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= other * nch) return;
  const int c = (int)(t % nch);
  const long long o = t / nch;
  auto at = [&](long long a) -> float * {
    return axis == 0 ? data + (o * sx + a) * nch + c : data + (a * sx + o) * nch + c;
  };
  if (m == 1) {
    for (long long i = 0; i < w; i++) if (i != lsz) *at(i) = *at(lsz);
    return;
  }
  const long long l0 = lsz - 1, r0 = lsz + m, lp = l0 + 1, rp = r0 - 1, l1 = -1, r1 = w;
  long long lt = l0, rt = r0, ls = 0, rs = 0, ds = 1;
  switch (bc) {
    case EU_BC_PERIODIC: ls = l0 + m; rs = r0 - m; ds = -1; break;
    case EU_BC_NATURAL:
    case EU_BC_MIRROR:   ls = l0 + 2; rs = r0 - 2; break;
    default:             ls = l0 + 1; rs = r0 - 1; break;     // CONSTANT, REFLECT (ZEROPAD: unused)
  }
  for (long long i = lsz > rsz ? lsz : rsz; i > 0; --i) {
    if (lt > l1) {
      if (bc == EU_BC_NATURAL) { const float a = *at(lp), b = *at(ls); *at(lt) = a + a - b; }
      else if (bc == EU_BC_CONSTANT) *at(lt) = *at(lp);
      else if (bc == EU_BC_ZEROPAD) *at(lt) = 0.0f;
      else *at(lt) = *at(ls);
      --lt; ls += ds;
    }
    if (rt < r1) {
      if (bc == EU_BC_NATURAL) { const float a = *at(rp), b = *at(rs); *at(rt) = a + a - b; }
      else if (bc == EU_BC_CONSTANT) *at(rt) = *at(rp);
      else if (bc == EU_BC_ZEROPAD) *at(rt) = 0.0f;
      else *at(rt) = *at(rs);
      ++rt; rs -= ds;
    }
  }
}

__device__ __forceinline__ float mirror_gate(float c, float lower, float upper)
{
  float cc = c - lower, w = upper - lower;
  cc = fabsf(cc);
  if (cc >= w) {
    float help = cc / (2 * w);
    help = truncf(help);
    help = help * (2 * w);
    float cm = cc - help;
    if (fabsf(cm) >= fabsf(2 * w)) cm = 0.0f;
    cm = cm - w;
    cm = fabsf(cm);
    cm = w - cm;
    cc = cm;
  }
  return cc + lower;
}

// fill_frame_t::eval, cubemap.h:724-809, for every frame pixel of one face
// fill_frame_t (cubemap.h:700-810) for frame pixel (x, y) of `face`'s section:
// the ray through the pixel, the cube face it hits, the pick-up coordinate in the
// IR, the bilinear sample from the IR as it stands; out[nch]
__device__ void fill_px(const float *ir, int nch, int face, long long x, long long y, long long S,
                        double refc_md, double model_to_px, float *out)
{
  int ishift = (int)S - 1, ithird = (int)(model_to_px * 2);
  int ix = (int)(2 * x) - ishift, iy = (int)(2 * y) - ishift;
  float c3[3];
  switch (face) {
    case 4: c3[0] = (float)ix;  c3[1] = (float)iy; c3[2] = (float)ithird; break;
    case 5: c3[0] = (float)-ix; c3[1] = (float)iy; c3[2] = (float)-ithird; break;
    case 1: c3[0] = (float)ithird;  c3[1] = (float)iy; c3[2] = (float)-ix; break;
    case 0: c3[0] = (float)-ithird; c3[1] = (float)iy; c3[2] = (float)ix; break;
    case 3: c3[0] = (float)-ix; c3[1] = (float)ithird;  c3[2] = (float)iy; break;
    default: c3[0] = (float)-ix; c3[1] = (float)-ithird; c3[2] = (float)-iy; break;
  }
  // ray_to_cubeface, geometry.h:1178-1289
  float ax = fabsf(c3[0]), ay = fabsf(c3[1]), az = fabsf(c3[2]);
  bool m1 = ax >= ay, m2 = ax >= az, m3 = ay >= az;
  int fv; float in0, in1;
  if (m1 && m2) { fv = c3[0] < 0.0f ? 0 : 1; in0 = -c3[2] / c3[0]; in1 = c3[1] / ax; }
  else if (!m2 && !m3) { fv = c3[2] < 0.0f ? 5 : 4; in0 = c3[0] / c3[2]; in1 = c3[1] / az; }
  else { fv = c3[1] < 0.0f ? 2 : 3; in0 = -c3[0] / ay; in1 = c3[2] / c3[1]; }
  // metrics_t::get_pickup_coordinate_px, cubemap.h:452-464 (double members)
  float p0 = (float)((double)in0 + refc_md), p1 = (float)((double)in1 + refc_md);
  p0 = p0 * (float)model_to_px;
  p1 = p1 * (float)model_to_px;
  p1 = p1 + (float)(fv * (int)S);
  p0 = p0 - .5f;
  p1 = p1 - .5f;
  // bilinear safe evaluator on the unfiltered IR (shift = 1 - degree):
  // REFLECT gates over (S, 6S), eval.h:1004-1059
  float gx = mirror_gate(p0, -0.5f, (float)((long double)(S - 1) + 0.5L));
  float gy = mirror_gate(p1, -0.5f, (float)((long double)(6 * S - 1) + 0.5L));
  float fx = floorf(gx), fy = floorf(gy);
  float tx = gx - fx, ty = gy - fy;
  const float *p = ir + ((long long)(int)fy * S + (long long)(int)fx) * nch;
  float wl0 = 1.0f - tx, wr0 = tx, wl1 = 1.0f - ty, wr1 = ty;
  for (int c = 0; c < nch; c++) {
    float sum = p[c] * wl0;
    sum = sum + p[nch + c] * wr0;
    sum = sum * wl1;
    float sub = p[S * nch + c] * wl0;
    sub = sub + p[S * nch + nch + c] * wr0;
    sum = sum + sub * wr1;
    out[c] = sum;
  }
}

// One of the four stripes of a section's frame, in the order fill_support works
// through them (cubemap.h:870-908): 0 above, 1 below, 2 left, 3 right of the cube
// face. The order is part of the result when the frame is one pixel wider on the
// right than on the left (odd face sizes): the frame pixels in the row below / the
// column right of the face then tie in the dominant-axis test and map onto their
// OWN face's edge, i.e. they read frame pixels of this section - among them their
// left / upper neighbour. `tie` excludes that row (stripe 1) / column (stripe 3)
// here; fill_tie_kernel does them in the reference's single-thread order.
__global__ void fill_frame_kernel(float *ir, int nch, int face, int stripe, long long F, long long S,
                                  long long lf, long long rf, double refc_md,
                                  double model_to_px, int tie)
{
  long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= S * S) return;
  long long x = t % S, y = t / S;
  bool mine;
  switch (stripe) {
    case 0: mine = y < lf; break;
    case 1: mine = y >= S - rf && !(tie && y == lf + F); break;
    case 2: mine = x < lf && y >= lf && y < S - rf; break;
    default: mine = x >= lf + F && y >= lf && y < S - rf && !(tie && x == lf + F); break;
  }
  if (!mine) return;
  float v[4];
  fill_px(ir, nch, face, x, y, S, refc_md, model_to_px, v);
  float *dst = ir + (((long long)face * S + y) * S + x) * nch;
  for (int c = 0; c < nch; c++) dst[c] = v[c];
}

// The tie row / column, ONE wavefront, in the order of zimt::process with one
// worker thread (wielding.h:337-463): stripe 1: the row y = lf + F in vectors of
// 16 pixels from x = 0, every vector evaluated from the array as it stands and
// then stored; stripe 3: the column x = lf + F row by row (it is the first lane
// of each row's vector, which reads the pixel above it).
__global__ void fill_tie_kernel(float *ir, int nch, int face, int stripe, long long F, long long S,
                                long long lf, long long rf, double refc_md, double model_to_px)
{
  const int lane = threadIdx.x;
  float *sec = ir + (long long)face * S * S * nch;
  if (stripe == 1) {
    const long long y = lf + F;
    for (long long x0 = 0; x0 < S; x0 += 16) {
      const long long x = x0 + lane;
      const bool on = lane < 16 && x < S;
      float v[4] = { 0.0f, 0.0f, 0.0f, 0.0f };
      if (on) fill_px(ir, nch, face, x, y, S, refc_md, model_to_px, v);
      // every lane's reads are done before any lane stores
      __builtin_amdgcn_s_waitcnt(0);
      __builtin_amdgcn_wave_barrier();
      if (on) for (int c = 0; c < nch; c++) sec[(y * S + x) * nch + c] = v[c];
      __threadfence();
      __builtin_amdgcn_s_waitcnt(0);
      __builtin_amdgcn_wave_barrier();
    }
  } else {
    const long long x = lf + F;
    for (long long y = lf; y < S - rf; y++) {
      float v[4] = { 0.0f, 0.0f, 0.0f, 0.0f };
      if (lane == 0) {
        fill_px(ir, nch, face, x, y, S, refc_md, model_to_px, v);
        for (int c = 0; c < nch; c++) sec[(y * S + x) * nch + c] = v[c];
      }
      __threadfence();
      __builtin_amdgcn_s_waitcnt(0);
    }
  }
}

inline unsigned blocks_for(long long n, int bs) { return (unsigned)((n + bs - 1) / bs); }

// frame of one axis: all slices at once, or - a core narrower than the frame - in zimt's order
void launch_brace(float *container, long long SX, long long SY, int nch, int axis, int bc, long long lsz,
                  long long rsz, hipStream_t st)
{
  const long long len = axis == 0 ? SX : SY, other = axis == 0 ? SY : SX, m = len - lsz - rsz;
  if (lsz + rsz <= 0) return;
  if (m != 1 && m < (lsz > rsz ? lsz : rsz) + 1) {
    hipLaunchKernelGGL(brace_seq_kernel, dim3(blocks_for(other * nch, 256)), dim3(256), 0, st, container, SX, SY,
                       nch, axis, bc, lsz, rsz);
    return;
  }
  hipLaunchKernelGGL(brace_kernel, dim3(blocks_for((lsz + rsz) * other * nch, 256)), dim3(256), 0, st, container,
                     SX, SY, nch, axis, bc, lsz, rsz);
}

// EU_HIP_IIR_STREAM=0: the one-thread-per-line kernels only (the form the streamed ones are
// checked against on the device, tests/test_gpu_prefilter_stream.py)
int iir_stream_on()   // bit 0: rows, bit 1: columns, bit 2: checkpoints + recomputation (3 passes per axis)
{
#ifdef EU_IIR_EXPERIMENT
  const char *d = getenv("EU_HIP_IIR_EXP");
  int dv = d ? atoi(d) : 0;
  (void)hipMemcpyToSymbol(HIP_SYMBOL(iir_exp), &dv, sizeof dv);
#endif
  const char *e = getenv("EU_HIP_IIR_STREAM");
  return e ? atoi(e) : 7;
}

// lines per wavefront: few, so that the lines of a large image make more wavefronts than the
// chip has SIMDs (1024) and a SIMD has two recursions to alternate between
int iir_env(const char *name, int dflt)
{
  const char *e = getenv(name);
  return e ? atoi(e) : dflt;
}

// Scratch for the checkpoints of the streamed sweeps (stream_causal): groups x blocks x 64 floats,
// stream-ordered allocation. Only when every pole's horizon lies inside the first block (iacc
// reads the causal result of that many samples) and EU_HIP_IIR_STREAM has bit 2 set.
float *iir_ckpt_alloc(const iir_dev &f, long long groups, int len, hipStream_t st)
{
  if (!(iir_stream_on() & 4) || len / 64 < 3) return nullptr;
  for (int k = 0; k < f.npoles; k++) if (f.horizon[k] >= 64) return nullptr;
  void *p = nullptr;
  if (hipMallocAsync(&p, (size_t)groups * (size_t)(len / 64) * 64 * sizeof(float), st) != hipSuccess) {
    (void)hipGetLastError();
    return nullptr;
  }
  return (float *)p;
}
void iir_ckpt_free(float *p, hipStream_t st) { if (p) (void)hipFreeAsync(p, st); }

// groups of lines -> workgroups of `wpg` independent wavefronts with `wave_lds` bytes each
template <class K, class... A>
void launch_stream(K kernel, unsigned groups, size_t wave_lds, hipStream_t st, A... args)
{
  // measured: 1, 2 and 4 wavefronts per workgroup take the same time (the sweeps are not bound by
  // where the wavefronts sit); 1 keeps the LDS request under 64 KB
  int wpg = iir_env("EU_HIP_IIR_WPG", 1);
  if (wpg < 1 || wpg > 4) wpg = 1;
  while (wpg > 1 && wpg * wave_lds > 160 * 1024) wpg--;
  const size_t lds = wpg * wave_lds;
  (void)hipFuncSetAttribute((const void *)kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  hipLaunchKernelGGL(kernel, dim3((groups + wpg - 1) / wpg), dim3(64 * wpg), lds, st, args..., groups);
}

template <int NCH, int R>
void launch_rows_nr(const iir_dev &f, float *base, unsigned groups, long long line_stride, int len, float *ck, hipStream_t st)
{
  launch_stream(filter_rows_stream_kernel<NCH, R>, groups,
                sizeof(float) * EU_IIR_BUFS * iir_bufs<tile_rows<NCH, R>>::BUF, st, f, base, line_stride, len, ck);
}

template <int NCH>
unsigned launch_rows_nch(int R, const iir_dev &f, float *base, long long nl, long long line_stride, int len, hipStream_t st)
{
  const unsigned groups = (unsigned)(nl / R);
  float *ck = iir_ckpt_alloc(f, groups, len, st);
  if (R == 4) launch_rows_nr<NCH, 4>(f, base, groups, line_stride, len, ck, st);
  else launch_rows_nr<NCH, 8>(f, base, groups, line_stride, len, ck, st);
  iir_ckpt_free(ck, st);
  return groups;
}

// nl rows of len samples x nch channels (es = nch), rows line_stride floats apart
void launch_filter_rows(const iir_dev &f, float *base, long long nl, int nch, long long line_stride,
                        int len, hipStream_t st)
{
  long long done = 0;
  const int R = iir_env("EU_HIP_IIR_ROWS", 8) == 4 ? 4 : 8;
  if ((iir_stream_on() & 1) && len >= 64 && nch >= 1 && nch <= 4 && nl >= R) {
    unsigned groups;
    switch (nch) {
      case 1: groups = launch_rows_nch<1>(R, f, base, nl, line_stride, len, st); break;
      case 2: groups = launch_rows_nch<2>(R, f, base, nl, line_stride, len, st); break;
      case 3: groups = launch_rows_nch<3>(R, f, base, nl, line_stride, len, st); break;
      default: groups = launch_rows_nch<4>(R, f, base, nl, line_stride, len, st);
    }
    done = (long long)groups * R;
  }
  if (done < nl)
    hipLaunchKernelGGL(filter_lines_kernel, dim3(blocks_for((nl - done) * nch, 64)), dim3(64), 0, st, f,
                       base + done * line_stride, nl - done, nch, line_stride, len, (long long)nch);
}

// nfloats adjacent columns (every float of a row is a line of its own), rows es floats apart
void launch_filter_cols(const iir_dev &f, float *base, long long nfloats, long long es, int len, hipStream_t st)
{
  long long done = 0;
  const int L = iir_env("EU_HIP_IIR_COLS", 32) == 16 ? 16 : 32;
  if ((iir_stream_on() & 2) && len >= 64 && nfloats >= L) {
    const unsigned groups = (unsigned)(nfloats / L);
    float *ck = iir_ckpt_alloc(f, groups, len, st);
    if (L == 16) launch_stream(filter_cols_stream_kernel<16>, groups, sizeof(float) * EU_IIR_BUFS * iir_bufs<tile_cols<16, false>>::BUF, st, f, base, es, len, ck);
    else launch_stream(filter_cols_stream_kernel<32>, groups, sizeof(float) * EU_IIR_BUFS * iir_bufs<tile_cols<32, false>>::BUF, st, f, base, es, len, ck);
    iir_ckpt_free(ck, st);
    done = (long long)groups * L;
  }
  if (done < nfloats)
    hipLaunchKernelGGL(filter_lines_kernel, dim3(blocks_for(nfloats - done, 64)), dim3(64), 0, st, f,
                       base + done, nfloats - done, 1, 1LL, len, es);
}

// environment.h:395-447: column t top -> bottom, then column t + down_off bottom -> top
void launch_filter_stacked(const iir_dev &f, float *core, long long nfloats, long long row_es, int H, hipStream_t st)
{
  long long done = 0;
  const int L = iir_env("EU_HIP_IIR_COLS", 32) == 16 ? 16 : 32;
  if ((iir_stream_on() & 2) && 2 * H >= 64 && nfloats >= L) {
    const unsigned groups = (unsigned)(nfloats / L);
    float *ck = iir_ckpt_alloc(f, groups, 2 * H, st);
    if (L == 16) launch_stream(filter_stacked_stream_kernel<16>, groups, sizeof(float) * EU_IIR_BUFS * iir_bufs<tile_cols<16, true>>::BUF, st, f, core, nfloats, row_es, H, ck);
    else launch_stream(filter_stacked_stream_kernel<32>, groups, sizeof(float) * EU_IIR_BUFS * iir_bufs<tile_cols<32, true>>::BUF, st, f, core, nfloats, row_es, H, ck);
    iir_ckpt_free(ck, st);
    done = (long long)groups * L;
  }
  if (done < nfloats)
    hipLaunchKernelGGL(filter_stacked_kernel, dim3(blocks_for(nfloats - done, 64)), dim3(64), 0, st, f,
                       core + done, nfloats - done, nfloats, row_es, H);
}

}  // namespace

extern "C" int eu_launch_prefilter(float *container, const eu_container *g, int nch, int bc0,
                                   int bc1, int degree, int spherical, void *stream)
{
  hipStream_t st = (hipStream_t)stream;
  const long long W = g->core[0], H = g->core[1], SX = g->shape[0], SY = g->shape[1];
  float *core = container + ((long long)g->left[1] * SX + g->left[0]) * nch;
  if (spherical) {
    // environment.h:356-522
    if (degree > 1) {
      iir_dev f = make_iir(EU_BC_PERIODIC, degree, 0.0001L, W);
      launch_filter_rows(f, core, H, nch, SX * nch, (int)W, st);
      iir_dev f2 = make_iir(EU_BC_PERIODIC, degree, 0.0001L, 2 * H);
      launch_filter_stacked(f2, core, (W / 2) * nch, SX * nch, (int)H, st);
    }
    long long n = (g->left[1] + g->right[1]) * W * nch;
    const long long vfr = g->left[1] > g->right[1] ? g->left[1] : g->right[1];
    if (n > 0 && H < vfr)
      hipLaunchKernelGGL(pole_rows_seq_kernel, dim3(blocks_for((W / 2) * nch, 64)), dim3(64), 0, st, core, W, H,
                         SX, nch, (long long)g->left[1], (long long)g->right[1]);
    else if (n > 0)
      hipLaunchKernelGGL(pole_rows_kernel, dim3(blocks_for(n, 256)), dim3(256), 0, st, core, W, H,
                         SX, nch, (long long)g->left[1], (long long)g->right[1]);
    // (an image narrower than its frame is braced slice by slice in zimt's order)
    launch_brace(container, SX, SY, nch, 0, bc0, (long long)g->left[0], (long long)g->right[0], st);
    return hipGetLastError() == hipSuccess ? 0 : -1;
  }
  // bspline::prefilter, zimt/bspline.h:1017-1041 + prefilter.h:133-190
  if (degree > 1) {
    iir_dev f0 = make_iir(bc0, degree, (long double)FLT_EPSILON, W);
    launch_filter_rows(f0, core, H, nch, SX * nch, (int)W, st);
    iir_dev f1 = make_iir(bc1, degree, (long double)FLT_EPSILON, H);
    launch_filter_cols(f1, core, W * nch, SX * nch, (int)H, st);
  }
  launch_brace(container, SX, SY, nch, 0, bc0, (long long)g->left[0], (long long)g->right[0], st);
  launch_brace(container, SX, SY, nch, 1, bc1, (long long)g->left[1], (long long)g->right[1], st);
  return hipGetLastError() == hipSuccess ? 0 : -1;
}

extern "C" int eu_launch_cubemap_build(const float *faces, float *ir, int nch, long F, long S,
                                       long lf, long rf, double refc_md, double model_to_px,
                                       int prefilter_degree, void *stream)
{
  hipStream_t st = (hipStream_t)stream;
  hipLaunchKernelGGL(place_faces_kernel, dim3(blocks_for(6LL * F * F * nch, 256)), dim3(256), 0,
                     st, faces, ir, nch, (long long)F, (long long)S, (long long)lf);
  if (lf || rf) {
    hipLaunchKernelGGL(mirror_around_kernel, dim3(blocks_for(6LL * 4 * (F + 2) * nch, 256)),
                       dim3(256), 0, st, ir, nch, (long long)F, (long long)S, (long long)lf,
                       (long long)rf);
    // the row below / column right of the face ties with the face's own edge when
    // 2 * (lf + F) - (S - 1) equals int(2 * model_to_px), i.e. rf == lf + 1
    const int tie = rf > 0 && 2 * (lf + F) - (S - 1) == (long)(int)(model_to_px * 2);
    for (int face = 0; face < 6; face++)
      for (int stripe = 0; stripe < 4; stripe++) {
        if ((stripe == 0 || stripe == 2) ? lf <= 0 : rf <= 0) continue;
        hipLaunchKernelGGL(fill_frame_kernel, dim3(blocks_for((long long)S * S, 256)), dim3(256), 0,
                           st, ir, nch, face, stripe, (long long)F, (long long)S, (long long)lf,
                           (long long)rf, refc_md, model_to_px, tie);
        if (tie && (stripe == 1 || stripe == 3))
          hipLaunchKernelGGL(fill_tie_kernel, dim3(1), dim3(64), 0, st, ir, nch, face, stripe,
                             (long long)F, (long long)S, (long long)lf, (long long)rf, refc_md,
                             model_to_px);
      }
  }
  if (prefilter_degree > 1) {
    // cubemap.h:921-946: per section, NATURAL x NATURAL, default tolerance
    iir_dev f = make_iir(EU_BC_NATURAL, prefilter_degree, (long double)FLT_EPSILON, S);
    launch_filter_rows(f, ir, 6LL * S, nch, (long long)S * nch, (int)S, st);
    for (int face = 0; face < 6; face++)
      launch_filter_cols(f, ir + (long long)face * S * S * nch, (long long)S * nch, (long long)S * nch, (int)S, st);
  }
  return hipGetLastError() == hipSuccess ? 0 : -1;
}

// ---------------------------------------------------------------------------
// x / c by q = x*rc; q' = fma(fma(-q, c, x), rc, q): verified for EVERY float x
// in [0, limit] against the correctly rounded quotient before the render kernel
// is allowed to use it for this constant (environment.h:993-1000 divides every
// source coordinate by float(extent))
// ---------------------------------------------------------------------------
__global__ void verify_const_div_kernel(float c, float rc, unsigned last_bits, int *bad)
{
  unsigned long long i = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x;
  const unsigned long long stride = (unsigned long long)gridDim.x * blockDim.x;
  int b = 0;
  for (; i <= last_bits; i += stride) {
    float x = eu_u2f((unsigned)i);
    float q = x * rc;
    float r = fmaf(-q, c, x);
    float q2 = fmaf(r, rc, q);
    b |= eu_f2u(q2) != eu_f2u(x / c);
  }
  if (b) atomicOr(bad, 1);
}

// returns 1 when the three-operation form is exact for all x in [0, limit]
extern "C" int eu_verify_const_div(float c, float limit, void *stream)
{
  if (!(c > 0.0f) || !(limit > 0.0f)) return 0;
  int *bad = nullptr;
  if (hipMalloc((void **)&bad, sizeof(int)) != hipSuccess) return 0;
  hipStream_t st = (hipStream_t)stream;
  (void)hipMemsetAsync(bad, 0, sizeof(int), st);
  unsigned last;
  memcpy(&last, &limit, 4);
  hipLaunchKernelGGL(verify_const_div_kernel, dim3(256 * 8), dim3(256), 0, st, c, 1.0f / c, last, bad);
  int h = 1;
  if (hipMemcpyAsync(&h, bad, sizeof(int), hipMemcpyDeviceToHost, st) != hipSuccess) h = 1;
  if (hipStreamSynchronize(st) != hipSuccess) h = 1;
  (void)hipFree(bad);
  return h ? 0 : 1;
}


// ---------------------------------------------------------------------------
// to_screen_t (envutil_payload.cc:251-413), the put stage of the tethered
// pipeline: every channel goes through lut_based_tf - in * 255.0f, clamp gate
// [0, 255] (NATURAL spline: eval.h:2096-2104), linear interpolation between two
// knots of the sRGB LUT (wl = 1 - t; s = c0 * wl; s += c1 * t), truncation to
// uint32 - and the four bytes are packed A<<24 | B<<16 | G<<8 | R. Runs over
// the float frame the render kernel just wrote (still in L2 / MALL for
// screen-sized frames).
// ---------------------------------------------------------------------------
__device__ __forceinline__ unsigned screen_channel(const float *__restrict__ lut, float v)
{
  float c = v * 255.0f;
  c = c < 0.0f ? 0.0f : c;
  c = c > 255.0f ? 255.0f : c;
  const float fl = floorf(c);
  const float t = c - fl;
  int i = (int)fl;
  i = min(max(i, 0), 255);            // NaN input (undefined in the reference): stay in the table
  const float wl = 1.0f - t;
  float s = lut[i] * wl;
  s = s + lut[i + 1] * t;
  return (unsigned)s;
}

__global__ __launch_bounds__(256) void to_screen_kernel(const float *__restrict__ in,
                                                        long long in_stride,
                                                        unsigned *__restrict__ out,
                                                        long long out_stride, int w, int nch,
                                                        const float *__restrict__ lut)
{
  const int x = blockIdx.x * 256 + threadIdx.x;
  const int y = blockIdx.y;
  if (x >= w) return;
  const float *px = in + (long long)y * in_stride + (long long)x * nch;
  const unsigned c1 = screen_channel(lut, px[0]);
  unsigned word;
  if (nch == 1) word = 0xFF000000u | (c1 << 16) | (c1 << 8) | c1;
  else if (nch == 2) word = (screen_channel(lut, px[1]) << 24) | (c1 << 16) | (c1 << 8) | c1;
  else {
    const unsigned c2 = screen_channel(lut, px[1]), c3 = screen_channel(lut, px[2]);
    const unsigned a = nch == 3 ? 0xFFu : screen_channel(lut, px[3]);
    word = (a << 24) | (c3 << 16) | (c2 << 8) | c1;
  }
  out[(long long)y * out_stride + x] = word;
}

extern "C" int eu_launch_to_screen(const float *in, long long in_stride, unsigned *out,
                                   long long out_stride, int w, int rows, int nch, const float *lut,
                                   void *stream)
{
  if (w <= 0 || rows <= 0) return 0;
  // rows sit in grid.y (at most 65535 per launch): taller frames go in slabs
  for (int r0 = 0; r0 < rows; r0 += 65535) {
    const int n = rows - r0 < 65535 ? rows - r0 : 65535;
    dim3 grid((unsigned)((w + 255) / 256), (unsigned)n);
    hipLaunchKernelGGL(to_screen_kernel, grid, dim3(256), 0, (hipStream_t)stream,
                       in + (long long)r0 * in_stride, in_stride, out + (long long)r0 * out_stride,
                       out_stride, w, nch, lut);
    if (hipGetLastError() != hipSuccess) return -1;
  }
  return 0;
}

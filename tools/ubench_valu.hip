// Micro-benchmark: issue rate of scalar vs packed fp32 VALU on gfx950.
// hipcc -O3 --offload-arch=gfx950 -ffp-contract=off tools/ubench_valu.hip -o /tmp/ubench && /tmp/ubench
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f2 __attribute__((ext_vector_type(2)));
#define N 4096
template <int MODE>
__global__ __launch_bounds__(256) void k(float *out, float a, float b)
{
  float x0 = threadIdx.x, x1 = x0 + 1, x2 = x0 + 2, x3 = x0 + 3, x4 = x0 + 4, x5 = x0 + 5, x6 = x0 + 6, x7 = x0 + 7;
  f2 p0 = {x0, x1}, p1 = {x2, x3}, p2 = {x4, x5}, p3 = {x6, x7}, pa = {a, a}, pb = {b, b};
  for (int i = 0; i < N; i++) {
    if (MODE == 0) {        // 8 scalar mul+add pairs (unfused)
      x0 = x0 * a + b; x1 = x1 * a + b; x2 = x2 * a + b; x3 = x3 * a + b;
      x4 = x4 * a + b; x5 = x5 * a + b; x6 = x6 * a + b; x7 = x7 * a + b;
      asm volatile("" : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7));
    } else if (MODE == 1) { // 4 packed mul+add pairs = same flops
      p0 = p0 * pa + pb; p1 = p1 * pa + pb; p2 = p2 * pa + pb; p3 = p3 * pa + pb;
      asm volatile("" : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3));
    } else if (MODE == 2) { // 8 rcp
      x0 = __builtin_amdgcn_rcpf(x0); x1 = __builtin_amdgcn_rcpf(x1); x2 = __builtin_amdgcn_rcpf(x2); x3 = __builtin_amdgcn_rcpf(x3);
      x4 = __builtin_amdgcn_rcpf(x4); x5 = __builtin_amdgcn_rcpf(x5); x6 = __builtin_amdgcn_rcpf(x6); x7 = __builtin_amdgcn_rcpf(x7);
      asm volatile("" : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7));
    } else if (MODE == 3) { // 8 true divisions
      x0 = x0 / a; x1 = x1 / a; x2 = x2 / a; x3 = x3 / a; x4 = x4 / a; x5 = x5 / a; x6 = x6 / a; x7 = x7 / a;
      asm volatile("" : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7));
    } else if (MODE == 4) { // 8 cndmask+cmp
      x0 = x0 < a ? b : x0; x1 = x1 < a ? b : x1; x2 = x2 < a ? b : x2; x3 = x3 < a ? b : x3;
      x4 = x4 < a ? b : x4; x5 = x5 < a ? b : x5; x6 = x6 < a ? b : x6; x7 = x7 < a ? b : x7;
      asm volatile("" : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7));
    } else if (MODE == 5) { // 8 sqrt (correctly rounded)
      x0 = sqrtf(x0); x1 = sqrtf(x1); x2 = sqrtf(x2); x3 = sqrtf(x3); x4 = sqrtf(x4); x5 = sqrtf(x5); x6 = sqrtf(x6); x7 = sqrtf(x7);
      asm volatile("" : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7));
    }
  }
  out[blockIdx.x * 256 + threadIdx.x] = x0 + x1 + x2 + x3 + x4 + x5 + x6 + x7 + p0.x + p0.y + p1.x + p1.y + p2.x + p2.y + p3.x + p3.y;
}
template <int MODE> void run(const char *name, float *d, int wavesPerSimd, double opsPerIter)
{
  int blocks = 256 * wavesPerSimd;   // 256 CUs, 4 waves per block = 1 per SIMD
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, d, 1.0001f, 0.5f);
  hipEventRecord(e0);
  hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, d, 1.0001f, 0.5f);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  double instr_per_simd = (double)N * opsPerIter * wavesPerSimd;
  printf("%-28s waves/SIMD %d: %.3f ms  -> %.2f ns per wave-instruction per SIMD (%.2f cycles @2.4GHz)\n",
         name, wavesPerSimd, ms, ms * 1e6 / instr_per_simd, ms * 1e6 / instr_per_simd * 2.4);
}
int main()
{
  float *d; hipMalloc(&d, 256 * 8 * 256 * 4);
  for (int w : {1, 2, 4, 8}) {
    run<0>("scalar mul,add (16 instr)", d, w, 16);
    run<1>("packed mul,add (8 instr)", d, w, 8);
    run<2>("v_rcp_f32 (8)", d, w, 8);
    run<3>("fp32 division (8)", d, w, 8);
    run<4>("cmp+cndmask (16)", d, w, 16);
    run<5>("sqrtf (8)", d, w, 8);
  }
  return 0;
}

// usage: hipcc -O2 --offload-arch=gfx950 tools/dma_align.hip -o /tmp/dma_align && /tmp/dma_align   (round 2: every 4-byte alignment, partial EXEC - all correct)
// LDS-DMA (global_load_lds_dwordx4) with global addresses of every 4-byte alignment: what lands in LDS?
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef __attribute__((address_space(3))) void *lvoid;
__device__ __forceinline__ void dma16(unsigned dst, const float *src)
{
  unsigned keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %2, off\n\ts_mov_b32 m0, %0"
               : "=&s"(keep) : "s"(dst), "v"(src) : "memory");
}
__global__ void k(const float *src, float *out, int off, int nlanes, int stride)
{
  __shared__ __attribute__((aligned(16))) float sm[1024];
  for (int i = threadIdx.x; i < 1024; i += 64) sm[i] = -1.0f;
  __syncthreads();
  unsigned lds = (unsigned)(unsigned long long)(lvoid)sm;
  int lane = threadIdx.x;
  if (lane < nlanes) {
    dma16(lds, src + off + (lane / 8) * stride + (lane % 8) * 4);
    dma16(lds + 1024, src + off + 4096 + (lane / 8) * stride + (lane % 8) * 4);
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  for (int i = threadIdx.x; i < 512; i += 64) out[i] = sm[i];
}
int main()
{
  const int N = 1 << 16;
  std::vector<float> h(N);
  for (int i = 0; i < N; i++) h[i] = (float)i;
  float *d, *o;
  hipMalloc(&d, N * 4); hipMalloc(&o, 512 * 4);
  hipMemcpy(d, h.data(), N * 4, hipMemcpyHostToDevice);
  int bad_total = 0;
  for (int stride : {32, 646, 1941})
    for (int nl : {64, 32, 16})
      for (int off = 0; off < 40; off++) {
        hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d, o, off, nl, stride);
        std::vector<float> r(512);
        hipMemcpy(r.data(), o, 512 * 4, hipMemcpyDeviceToHost);
        int bad = 0, first = -1;
        for (int l = 0; l < nl; l++)
          for (int j = 0; j < 4; j++) {
            float e0 = (float)(off + (l / 8) * stride + (l % 8) * 4 + j);
            if (r[l * 4 + j] != e0) { bad++; if (first < 0) first = l * 4 + j; }
            if (r[256 + l * 4 + j] != e0 + 4096) { bad++; if (first < 0) first = 256 + l * 4 + j; }
          }
        if (bad) { printf("stride %d lanes %d off %d: %d bad, first at %d got %g\n", stride, nl, off, bad, first, r[first]); bad_total += bad; }
      }
  printf("total bad %d\n", bad_total);
  return 0;
}

// Calibration of rocprofv3's FETCH_SIZE for the access shape of eu_render5_kernel's staging (MI355X_MICROARCH.md, HBM:
// "Other access widths are uncalibrated: calibrate on a known byte count in your own access pattern"): LDS-DMA,
// 16 bytes per lane at a 12-byte stride (an RGB texel plus one float), rows of `bw` texels, `k = 64 / bw` rows per
// instruction - against the same bytes moved by a plain coalesced 16-byte-per-lane streaming read (the case the guide
// calibrated: FETCH_SIZE = 1/2 of the bytes). Every byte of the buffer (1.61 GB, the headline source's size) is read
// exactly once by each kernel.
//   hipcc -O3 --offload-arch=gfx950 tools/calib_fetch.hip -o /tmp/calib_fetch
//   rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d out -- /tmp/calib_fetch
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

typedef __attribute__((address_space(3))) void *lds_void;
typedef const __attribute__((address_space(1))) void *gbl_void;

// box rows of BW texels (12 bytes each), BH rows; boxes tile the image without overlap
template <int BW, int BH>
__global__ __launch_bounds__(256) void dma_boxes(const float *src, long long pitch_floats, int boxes_x, int boxes_y, float *sink)
{
  __shared__ __attribute__((aligned(16))) float lds[4][BW * BH * 4 + 64];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  constexpr int K = 64 / BW;
  const int r = lane / BW, c = lane % BW;
  float acc = 0.0f;
  for (long long b = (long long)blockIdx.x * 4 + wave; b < (long long)boxes_x * boxes_y; b += (long long)gridDim.x * 4) {
    const int by = (int)(b / boxes_x), bx = (int)(b % boxes_x);
    const float *base = src + (long long)by * BH * pitch_floats + (long long)bx * BW * 3;
    if (r < K)
      for (int row = 0; row + K <= BH; row += K)
        __builtin_amdgcn_global_load_lds((gbl_void)(base + (long long)(row + r) * pitch_floats + c * 3),
                                         (lds_void)(&lds[wave][(row * BW) * 4]), 16, 0, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    acc += lds[wave][lane * 4];
  }
  if (acc == 12345.678f) sink[0] = acc;
}

__global__ __launch_bounds__(256) void stream16(const float4 *src, long long n, float *sink)
{
  float acc = 0.0f;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) {
    const float4 v = src[i];
    acc += v.x + v.w;
  }
  if (acc == 12345.678f) sink[0] = acc;
}

int main()
{
  const int W = 16389, H = 8196;                       // the headline source's braced container
  const long long pitch = (long long)W * 3;
  const long long nfloats = pitch * H;
  float *d, *sink;
  hipMalloc(&d, nfloats * 4 + 4096); hipMalloc(&sink, 16);
  hipMemset(d, 0, nfloats * 4 + 4096);
  constexpr int BW = 21, BH = 15;                      // a typical equatorial box of a 16x8 tile
  const int bx = W / BW, by = H / BH;
  const double box_bytes = (double)bx * by * BW * BH * 12.0;
  for (int rep = 0; rep < 3; rep++) {
    hipLaunchKernelGGL((dma_boxes<BW, BH>), dim3(4096), dim3(256), 0, 0, d, pitch, bx, by, sink);
    hipLaunchKernelGGL(stream16, dim3(4096), dim3(256), 0, 0, (const float4 *)d, nfloats / 4, sink);
  }
  hipDeviceSynchronize();
  printf("dma_boxes<%d,%d>: %.0f bytes of texels read once (%.3f GB); stream16: %.0f bytes (%.3f GB)\n", BW, BH, box_bytes,
         box_bytes / 1e9, (double)(nfloats / 4) * 16, (double)(nfloats / 4) * 16 / 1e9);
  return 0;
}

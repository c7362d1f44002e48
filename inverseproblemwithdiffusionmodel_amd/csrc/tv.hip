// Total-variation regulariser of the TV baseline (reference: scripts/acdc_SENSE_TV.py:76-83 -- kornia's TotalVariation on the
// complex image inside ncsn/models/MAP_optimizers.py:41-48 MAPModel.forward):
//   TV(x) = sum_{i,j} |x[i+1,j] - x[i,j]| + sum_{i,j} |x[i,j+1] - x[i,j]|            (complex modulus, per image)
// and the gradient torch's autograd hands the optimiser for the complex parameter (d|z| -> z / |z|, 0 at z = 0):
//   g[i,j] = s(x[i,j] - x[i-1,j]) - s(x[i+1,j] - x[i,j]) + s(x[i,j] - x[i,j-1]) - s(x[i,j+1] - x[i,j]),   s(z) = z / |z|.
// Both HBM-bound single passes (8 B read + 8 B written per pixel; neighbours come from L1 / L2).
#include "ipdm_common.h"

namespace {

__device__ __forceinline__ float2 unit_of(float2 d) {
  const float n = sqrtf(d.x * d.x + d.y * d.y);
  return n > 0.f ? make_float2(d.x / n, d.y / n) : make_float2(0.f, 0.f);
}

__global__ __launch_bounds__(256) void tv_grad_kernel(const float2* __restrict__ x, float2* __restrict__ g, int H, int W,
                                                      int64_t total) {
  const int64_t HW = (int64_t)H * W;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    const int64_t p = i % HW;
    const int r = (int)(p / W), c = (int)(p % W);
    const float2 v = x[i];
    float2 acc = make_float2(0.f, 0.f);
    if (r > 0) { const float2 q = x[i - W]; const float2 u = unit_of(make_float2(v.x - q.x, v.y - q.y)); acc.x += u.x; acc.y += u.y; }
    if (r + 1 < H) { const float2 q = x[i + W]; const float2 u = unit_of(make_float2(q.x - v.x, q.y - v.y)); acc.x -= u.x; acc.y -= u.y; }
    if (c > 0) { const float2 q = x[i - 1]; const float2 u = unit_of(make_float2(v.x - q.x, v.y - q.y)); acc.x += u.x; acc.y += u.y; }
    if (c + 1 < W) { const float2 q = x[i + 1]; const float2 u = unit_of(make_float2(q.x - v.x, q.y - v.y)); acc.x -= u.x; acc.y -= u.y; }
    g[i] = acc;
  }
}

// one workgroup per image, float64 accumulation, fixed reduction order
__global__ __launch_bounds__(256) void tv_value_kernel(const float2* __restrict__ x, double* __restrict__ out, int H, int W) {
  __shared__ double red[256];
  const float2* img = x + (int64_t)blockIdx.x * H * W;
  double acc = 0.0;
  for (int p = threadIdx.x; p < H * W; p += 256) {
    const int r = p / W, c = p % W;
    const float2 v = img[p];
    if (r + 1 < H) { const float2 q = img[p + W]; acc += sqrt((double)(q.x - v.x) * (q.x - v.x) + (double)(q.y - v.y) * (q.y - v.y)); }
    if (c + 1 < W) { const float2 q = img[p + 1]; acc += sqrt((double)(q.x - v.x) * (q.x - v.x) + (double)(q.y - v.y) * (q.y - v.y)); }
  }
  red[threadIdx.x] = acc;
  __syncthreads();
  for (int st = 128; st > 0; st >>= 1) {
    if ((int)threadIdx.x < st) red[threadIdx.x] += red[threadIdx.x + st];
    __syncthreads();
  }
  if (threadIdx.x == 0) out[blockIdx.x] = red[0];
}

}  // namespace

extern "C" int ipdm_tv_grad_c64(const float* x, float* g, int n_images, int H, int W, void* stream) {
  IPDM_REQUIRE(n_images >= 0 && H > 0 && W > 0);
  if (n_images == 0) return IPDM_OK;
  IPDM_REQUIRE(x && g && x != g);
  const int64_t total = (int64_t)n_images * H * W;
  hipLaunchKernelGGL(tv_grad_kernel, dim3(ipdm_ew_grid(total, 256)), dim3(256), 0, ipdm_stream(stream),
                     reinterpret_cast<const float2*>(x), reinterpret_cast<float2*>(g), H, W, (long long)total);
  return ipdm_launch_status();
}

extern "C" int ipdm_tv_c64(const float* x, double* out, int n_images, int H, int W, void* stream) {
  IPDM_REQUIRE(n_images >= 0 && H > 0 && W > 0);
  if (n_images == 0) return IPDM_OK;
  IPDM_REQUIRE(x && out);
  hipLaunchKernelGGL(tv_value_kernel, dim3(n_images), dim3(256), 0, ipdm_stream(stream), reinterpret_cast<const float2*>(x), out,
                     H, W);
  return ipdm_launch_status();
}

// Per-image max |x| of an activation tensor as a maxima vector ([B][IPDM_AMAX_SLOT], ipdm.h): the input of the f16x2
// convolutions' dynamic range (conv_kernel.h, hx_dynamic_scale) where no producer handed the maxima over.  One HBM pass (4 B per
// element, float4 loads), a wave reduction and one atomic max per wave on a way of the image's slot.  The vector is zeroed by the same
// call (hipMemsetAsync: a memset node under graph capture).  NaN inputs are ignored by the max (the convolution then produces
// NaN outputs from them as any fp32 convolution would).
#include "ipdm_common.h"

namespace {

__global__ __launch_bounds__(256) void absmax_kernel(const float* __restrict__ x, unsigned* __restrict__ amax, int64_t per_image,
                                                     int blocks_per_image) {
  const int img = blockIdx.x / blocks_per_image, blk = blockIdx.x % blocks_per_image;
  const float* p = x + (int64_t)img * per_image;
  const int64_t n4 = ((reinterpret_cast<uintptr_t>(p) & 15) == 0) ? per_image / 4 : 0;
  const int64_t stride = (int64_t)blocks_per_image * 256;
  float m0 = 0.f, m1 = 0.f, m2 = 0.f, m3 = 0.f;               // four loads in flight per lane and trip
  int64_t i = (int64_t)blk * 256 + threadIdx.x;
  auto am4 = [](float4 v) { return fmaxf(fmaxf(fabsf(v.x), fabsf(v.y)), fmaxf(fabsf(v.z), fabsf(v.w))); };
  const float4* p4 = reinterpret_cast<const float4*>(p);
  for (; i + 3 * stride < n4; i += 4 * stride) {
    const float4 a = p4[i], b = p4[i + stride], c = p4[i + 2 * stride], d = p4[i + 3 * stride];
    m0 = fmaxf(m0, am4(a)); m1 = fmaxf(m1, am4(b)); m2 = fmaxf(m2, am4(c)); m3 = fmaxf(m3, am4(d));
  }
  for (; i < n4; i += stride) m0 = fmaxf(m0, am4(p4[i]));
  float m = fmaxf(fmaxf(m0, m1), fmaxf(m2, m3));
  for (int64_t k = n4 * 4 + (int64_t)blk * 256 + threadIdx.x; k < per_image; k += stride) m = fmaxf(m, fabsf(p[k]));
  ipdm_amax_commit(m, reinterpret_cast<float*>(amax) + (size_t)img * IPDM_AMAX_SLOT, (int)(blk * 4 + (threadIdx.x >> 6)));
}

}  // namespace

extern "C" int ipdm_absmax_f32(const float* x, float* amax, int n_images, int64_t per_image, void* stream) {
  IPDM_REQUIRE(n_images >= 0 && per_image >= 0);
  if (n_images == 0) return IPDM_OK;
  IPDM_REQUIRE(amax && (x || per_image == 0));
  hipStream_t s = ipdm_stream(stream);
  hipError_t e = hipMemsetAsync(amax, 0, (size_t)n_images * IPDM_AMAX_SLOT * sizeof(float), s);
  if (e != hipSuccess) return (int)e;
  if (per_image == 0) return IPDM_OK;
  int64_t bpi = (per_image / 4 + 255) / 256;                  // one float4 per thread and trip ...
  const int64_t cap = (2048 + n_images - 1) / n_images;       // ... capped at ~2048 workgroups in all
  bpi = bpi < 1 ? 1 : (bpi > cap ? cap : bpi);
  hipLaunchKernelGGL(absmax_kernel, dim3((unsigned)(bpi * n_images)), dim3(256), 0, s, x, reinterpret_cast<unsigned*>(amax),
                     (long long)per_image, (int)bpi);
  return ipdm_launch_status();
}

// On-device reporting for many-sample studies (reference: helpers/metrics.py:21-102 -- scikit-image
// normalized_root_mse('euclidean') / structural_similarity, numpy mean / std over posterior samples --,
// helpers/visualizations.py:93,117,121: metrics are taken on MAGNITUDE images, posterior panels are mean / std of |x| and of
// angle(x)).  Everything here is a deterministic reduction: fixed loop orders, no atomics, float64 accumulators (the
// reference computes these in float64 numpy), so a metric does not depend on launch geometry.
#include "ipdm_common.h"

namespace {

__device__ __forceinline__ double block_sum_1024(double v, double* scratch) {
  v = ipdm_wave_sum(v);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  __syncthreads();
  if (lane == 0) scratch[wave] = v;
  __syncthreads();
  double t = 0.0;
  for (int w = 0; w < (int)(blockDim.x >> 6); ++w) t += scratch[w];      // same order in every thread
  return t;
}

// |x| of complex64 -> float32
__global__ __launch_bounds__(256) void magnitude_kernel(const float2* __restrict__ x, float* __restrict__ out, int64_t n) {
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
    const float2 v = x[i];
    out[i] = hypotf(v.x, v.y);
  }
}

// samples [n][HW] complex64 -> planes [7][HW] float64: sum |x|, sum |x|^2, sum angle, sum angle^2, sum Re, sum Im,
// sum |angle| -- the reference's phase "std" is np.std(np.abs(angle)) (helpers/metrics.py:82-83 applied to the phase stack)
// (sample order 0..n-1 per pixel: the partial sums of a shard are what sharding.all_reduce_posterior adds up; float64
//  sums, because std = sqrt(E[x^2] - E[x]^2) of angles of O(pi) loses 1e-4 in float32)
__global__ __launch_bounds__(256) void posterior_moments_kernel(const float2* __restrict__ s, double* __restrict__ planes,
                                                                int n, int64_t HW) {
  for (int64_t p = (int64_t)blockIdx.x * 256 + threadIdx.x; p < HW; p += (int64_t)gridDim.x * 256) {
    double m1 = 0, m2 = 0, a1 = 0, a2 = 0, re = 0, im = 0, aa = 0;
    for (int k = 0; k < n; ++k) {
      const float2 v = s[(size_t)k * HW + p];
      const double mag = hypotf(v.x, v.y), ang = atan2f(v.y, v.x);      // float32 |x| and angle, as numpy on complex64
      m1 += mag; m2 += mag * mag; a1 += ang; a2 += ang * ang; re += v.x; im += v.y; aa += fabs(ang);
    }
    planes[p] = m1; planes[HW + p] = m2; planes[2 * HW + p] = a1; planes[3 * HW + p] = a2;
    planes[4 * HW + p] = re; planes[5 * HW + p] = im; planes[6 * HW + p] = aa;
  }
}

// out[i] = sqrt(sum (a - b)^2 / sum a^2): skimage normalized_root_mse(image_true = a, image_test = b, 'euclidean');
// the reference passes the RECONSTRUCTION as a (helpers/metrics.py:72)
__global__ __launch_bounds__(1024) void nrmse_kernel(const float* __restrict__ a, const float* __restrict__ b,
                                                     double* __restrict__ out, int64_t elems, int64_t b_stride) {
  __shared__ double scratch[16];
  const float* ai = a + (size_t)blockIdx.x * elems;
  const float* bi = b + (size_t)blockIdx.x * b_stride;
  double num = 0.0, den = 0.0;
  for (int64_t i = threadIdx.x; i < elems; i += 1024) {
    const double x = ai[i], d = x - (double)bi[i];
    num += d * d;
    den += x * x;
  }
  num = block_sum_1024(num, scratch);
  den = block_sum_1024(den, scratch);
  if (threadIdx.x == 0) out[blockIdx.x] = sqrt(num / den);
}

// mean structural similarity of two [H][W] images, skimage defaults: 7x7 uniform window, K1 = 0.01, K2 = 0.03, sample
// covariance (NP / (NP - 1)), mean over the pixels whose window lies inside the image
__global__ __launch_bounds__(1024) void ssim_kernel(const float* __restrict__ a, const float* __restrict__ b,
                                                    double* __restrict__ out, int H, int W, int64_t b_stride,
                                                    double data_range) {
  constexpr int WIN = 7;
  __shared__ double scratch[16];
  const float* ai = a + (size_t)blockIdx.x * H * W;
  const float* bi = b + (size_t)blockIdx.x * b_stride;
  const int oh = H - (WIN - 1), ow = W - (WIN - 1);
  const double NP = WIN * WIN, cov_norm = NP / (NP - 1.0);
  const double C1 = (0.01 * data_range) * (0.01 * data_range), C2 = (0.03 * data_range) * (0.03 * data_range);
  double acc = 0.0;
  for (int p = threadIdx.x; p < oh * ow; p += 1024) {
    const int oy = p / ow, ox = p - oy * ow;
    double sa = 0, sb = 0, saa = 0, sbb = 0, sab = 0;
    for (int dy = 0; dy < WIN; ++dy) {
      const float* ra = ai + (size_t)(oy + dy) * W + ox;
      const float* rb = bi + (size_t)(oy + dy) * W + ox;
#pragma unroll
      for (int dx = 0; dx < WIN; ++dx) {
        const double x = ra[dx], y = rb[dx];
        sa += x; sb += y; saa += x * x; sbb += y * y; sab += x * y;
      }
    }
    const double ux = sa / NP, uy = sb / NP;
    const double vx = cov_norm * (saa / NP - ux * ux), vy = cov_norm * (sbb / NP - uy * uy);
    const double vxy = cov_norm * (sab / NP - ux * uy);
    acc += ((2 * ux * uy + C1) * (2 * vxy + C2)) / ((ux * ux + uy * uy + C1) * (vx + vy + C2));
  }
  acc = block_sum_1024(acc, scratch);
  if (threadIdx.x == 0) out[blockIdx.x] = acc / ((double)oh * ow);
}

}  // namespace

extern "C" int ipdm_magnitude_c64(const float* x, float* out, int64_t n, void* stream) {
  IPDM_REQUIRE(n >= 0);
  if (n == 0) return IPDM_OK;
  IPDM_REQUIRE(x && out);
  hipLaunchKernelGGL(magnitude_kernel, dim3(ipdm_ew_grid(n, 256)), dim3(256), 0, ipdm_stream(stream),
                     reinterpret_cast<const float2*>(x), out, (long long)n);
  return ipdm_launch_status();
}

extern "C" int ipdm_posterior_moments_c64(const float* samples, double* planes, int n_samples, int64_t HW, void* stream) {
  IPDM_REQUIRE(n_samples >= 0 && HW >= 0);
  if (HW == 0) return IPDM_OK;
  IPDM_REQUIRE(planes && (samples || n_samples == 0));
  hipLaunchKernelGGL(posterior_moments_kernel, dim3(ipdm_ew_grid(HW, 256)), dim3(256), 0, ipdm_stream(stream),
                     reinterpret_cast<const float2*>(samples), planes, n_samples, (long long)HW);
  return ipdm_launch_status();
}

extern "C" int ipdm_nrmse_f32(const float* img, const float* ref, double* out, int n_images, int64_t elems,
                              int ref_broadcast, void* stream) {
  IPDM_REQUIRE(n_images >= 0 && elems > 0);
  if (n_images == 0) return IPDM_OK;
  IPDM_REQUIRE(img && ref && out);
  hipLaunchKernelGGL(nrmse_kernel, dim3(n_images), dim3(1024), 0, ipdm_stream(stream), img, ref, out, (long long)elems,
                     (long long)(ref_broadcast ? 0 : elems));
  return ipdm_launch_status();
}

extern "C" int ipdm_ssim_f32(const float* img, const float* ref, double* out, int n_images, int H, int W,
                             int ref_broadcast, double data_range, void* stream) {
  IPDM_REQUIRE(n_images >= 0 && H >= 7 && W >= 7 && data_range > 0);
  if (n_images == 0) return IPDM_OK;
  IPDM_REQUIRE(img && ref && out);
  hipLaunchKernelGGL(ssim_kernel, dim3(n_images), dim3(1024), 0, ipdm_stream(stream), img, ref, out, H, W,
                     (long long)(ref_broadcast ? 0 : (int64_t)H * W), data_range);
  return ipdm_launch_status();
}

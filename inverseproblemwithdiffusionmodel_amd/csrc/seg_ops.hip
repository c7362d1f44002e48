// Building blocks of the segmentation-likelihood guidance (reference: ncsn/models/__init__.py:197-215
// compute_seg_grad = d/dX sum log softmax(seg(X))[label] through a MONAI UNet, ALD_optimizers.py:272-286 adjust_grad).
// The UNet's stride-2 convolutions / transposed convolutions and their input-gradients all reduce to the stride-1 MFMA
// convolution kernels between a zero-insertion and an even-position subsampling; what is new here is the glue:
// zero-insert / subsample, InstanceNorm + PReLU forward and backward (one workgroup per plane, two deterministic block
// reductions), the log-likelihood gradient at the logits, and the schedule-scaled accumulate into the score.
#include "ipdm_common.h"

namespace {

__device__ __forceinline__ float block_sum_f(float v, float* scratch) {
  v = ipdm_wave_sum(v);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  __syncthreads();
  if (lane == 0) scratch[wave] = v;
  __syncthreads();
  float t = 0.f;
  for (int w = 0; w < (int)(blockDim.x >> 6); ++w) t += scratch[w];
  return t;
}

// out [P][2H][2W]: out[2i][2j] = x[i][j], zero elsewhere
__global__ __launch_bounds__(256) void zero_insert2_kernel(const float* __restrict__ x, float* __restrict__ out, int64_t n_out,
                                                           int H2, int W2) {
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n_out; i += (int64_t)gridDim.x * 256) {
    const int c = (int)(i % W2);
    const int64_t t = i / W2;
    const int r = (int)(t % H2);
    const int64_t p = t / H2;
    float v = 0.f;
    if (((r | c) & 1) == 0) v = x[(p * (H2 >> 1) + (r >> 1)) * (int64_t)(W2 >> 1) + (c >> 1)];
    out[i] = v;
  }
}

// out [P][Ho][Wo] = x[2i + oy][2j + ox]     (x [P][H][W]; the caller guarantees 2(Ho-1)+oy < H, 2(Wo-1)+ox < W)
__global__ __launch_bounds__(256) void subsample2_kernel(const float* __restrict__ x, float* __restrict__ out, int64_t n_out,
                                                         int H, int W, int Ho, int Wo, int oy, int ox) {
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n_out; i += (int64_t)gridDim.x * 256) {
    const int c = (int)(i % Wo);
    const int64_t t = i / Wo;
    const int r = (int)(t % Ho);
    const int64_t p = t / Ho;
    out[i] = x[(p * H + 2 * r + oy) * (int64_t)W + 2 * c + ox];
  }
}

// plane-wise InstanceNorm (biased variance, eps inside the sqrt, no affine) followed by PReLU with ONE learned slope:
//   xhat = (x - mean) * rstd ; y = xhat > 0 ? xhat : slope * xhat        one workgroup per (b, c) plane
__global__ __launch_bounds__(256) void in_prelu_fwd_kernel(const float* __restrict__ x, const float* __restrict__ slope,
                                                           float* __restrict__ xhat, float* __restrict__ y,
                                                           float* __restrict__ rstd_out, int HW, float eps) {
  __shared__ float scratch[4];
  const float* xp = x + (size_t)blockIdx.x * HW;
  float s = 0.f;
  for (int i = threadIdx.x; i < HW; i += 256) s += xp[i];
  const float mean = block_sum_f(s, scratch) / (float)HW;
  float q = 0.f;
  for (int i = threadIdx.x; i < HW; i += 256) {
    const float d = xp[i] - mean;
    q += d * d;
  }
  const float rstd = rsqrtf(block_sum_f(q, scratch) / (float)HW + eps);
  const float a = slope ? slope[0] : 1.f;                  // NULL: no activation (conv_only layers never come here)
  for (int i = threadIdx.x; i < HW; i += 256) {
    const float h = (xp[i] - mean) * rstd;
    xhat[(size_t)blockIdx.x * HW + i] = h;
    y[(size_t)blockIdx.x * HW + i] = h > 0.f ? h : a * h;
  }
  if (threadIdx.x == 0) rstd_out[blockIdx.x] = rstd;
}

// input-gradient of the above: g_h = g_y * (xhat > 0 ? 1 : slope) ; g_x = rstd * (g_h - mean(g_h) - xhat * mean(g_h * xhat))
__global__ __launch_bounds__(256) void in_prelu_bwd_kernel(const float* __restrict__ gy, const float* __restrict__ xhat,
                                                           const float* __restrict__ rstd, const float* __restrict__ slope,
                                                           float* __restrict__ gx, int HW) {
  __shared__ float scratch[4];
  const size_t base = (size_t)blockIdx.x * HW;
  const float a = slope ? slope[0] : 1.f;
  float s1 = 0.f, s2 = 0.f;
  for (int i = threadIdx.x; i < HW; i += 256) {
    const float h = xhat[base + i];
    const float g = gy[base + i] * (h > 0.f ? 1.f : a);
    s1 += g;
    s2 += g * h;
  }
  const float m1 = block_sum_f(s1, scratch) / (float)HW;
  const float m2 = block_sum_f(s2, scratch) / (float)HW;
  const float r = rstd[blockIdx.x];
  for (int i = threadIdx.x; i < HW; i += 256) {
    const float h = xhat[base + i];
    const float g = gy[base + i] * (h > 0.f ? 1.f : a);
    gx[base + i] = r * (g - m1 - h * m2);
  }
}

// d/dlogits of sum_pixels log softmax(logits)[label]:  g[c] = [c == label] - softmax(logits)[c]     logits [B][C][HW]
__global__ __launch_bounds__(256) void seg_loglh_grad_kernel(const float* __restrict__ logits, const int64_t* __restrict__ label,
                                                             float* __restrict__ g, int B, int C, int64_t HW) {
  const int64_t n = (int64_t)B * HW;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
    const int64_t b = i / HW, p = i - b * HW;
    const float* lp = logits + (size_t)b * C * HW + p;
    float mx = lp[0];
    for (int c = 1; c < C; ++c) mx = fmaxf(mx, lp[(size_t)c * HW]);
    float den = 0.f;
    for (int c = 0; c < C; ++c) den += expf(lp[(size_t)c * HW] - mx);
    const int64_t lab = label[i];
    for (int c = 0; c < C; ++c)
      g[(size_t)b * C * HW + (size_t)c * HW + p] = (c == lab ? 1.f : 0.f) - expf(lp[(size_t)c * HW] - mx) / den;
  }
}

// y += scale * x * (mask ? mask : 1), scale = seg_scale of the device schedule (or the host scalar)
__global__ __launch_bounds__(256) void axpy_sched_kernel(float* __restrict__ y, const float* __restrict__ x,
                                                         const int64_t* __restrict__ mask, int64_t mask_period,
                                                         const ipdm_sched_t* __restrict__ sched, float scale, int64_t n) {
  if (sched) scale = sched->seg_scale;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
    float v = x[i];
    if (mask) v *= (float)mask[i % mask_period];
    y[i] = y[i] + v * scale;
  }
}

}  // namespace

extern "C" int ipdm_zero_insert2_f32(const float* x, float* out, int planes, int H, int W, void* stream) {
  IPDM_REQUIRE(planes >= 0 && H > 0 && W > 0);
  if (planes == 0) return IPDM_OK;
  IPDM_REQUIRE(x && out);
  const int64_t n = (int64_t)planes * 4 * H * W;
  hipLaunchKernelGGL(zero_insert2_kernel, dim3(ipdm_ew_grid(n, 256)), dim3(256), 0, ipdm_stream(stream), x, out, (long long)n,
                     2 * H, 2 * W);
  return ipdm_launch_status();
}

extern "C" int ipdm_subsample2_f32(const float* x, float* out, int planes, int H, int W, int oy, int ox, int Ho, int Wo,
                                   void* stream) {
  IPDM_REQUIRE(planes >= 0 && H > 0 && W > 0 && Ho > 0 && Wo > 0 && oy >= 0 && ox >= 0);
  IPDM_REQUIRE(2 * (Ho - 1) + oy < H && 2 * (Wo - 1) + ox < W);
  if (planes == 0) return IPDM_OK;
  IPDM_REQUIRE(x && out);
  const int64_t n = (int64_t)planes * Ho * Wo;
  hipLaunchKernelGGL(subsample2_kernel, dim3(ipdm_ew_grid(n, 256)), dim3(256), 0, ipdm_stream(stream), x, out, (long long)n,
                     H, W, Ho, Wo, oy, ox);
  return ipdm_launch_status();
}

extern "C" int ipdm_in_prelu_fwd_f32(const float* x, const float* slope, float* xhat, float* y, float* rstd, int planes,
                                     int HW, float eps, void* stream) {
  IPDM_REQUIRE(planes >= 0 && HW > 0);
  if (planes == 0) return IPDM_OK;
  IPDM_REQUIRE(x && xhat && y && rstd);
  hipLaunchKernelGGL(in_prelu_fwd_kernel, dim3(planes), dim3(256), 0, ipdm_stream(stream), x, slope, xhat, y, rstd, HW, eps);
  return ipdm_launch_status();
}

extern "C" int ipdm_in_prelu_bwd_f32(const float* gy, const float* xhat, const float* rstd, const float* slope, float* gx,
                                     int planes, int HW, void* stream) {
  IPDM_REQUIRE(planes >= 0 && HW > 0);
  if (planes == 0) return IPDM_OK;
  IPDM_REQUIRE(gy && xhat && rstd && gx);
  hipLaunchKernelGGL(in_prelu_bwd_kernel, dim3(planes), dim3(256), 0, ipdm_stream(stream), gy, xhat, rstd, slope, gx, HW);
  return ipdm_launch_status();
}

extern "C" int ipdm_seg_loglh_grad_f32(const float* logits, const int64_t* label, float* g, int B, int C, int64_t HW,
                                       void* stream) {
  IPDM_REQUIRE(B >= 0 && C > 0 && HW > 0);
  if (B == 0) return IPDM_OK;
  IPDM_REQUIRE(logits && label && g);
  hipLaunchKernelGGL(seg_loglh_grad_kernel, dim3(ipdm_ew_grid((int64_t)B * HW, 256)), dim3(256), 0, ipdm_stream(stream), logits,
                     reinterpret_cast<const long*>(label), g, B, C, (long long)HW);
  return ipdm_launch_status();
}

extern "C" int ipdm_axpy_sched_f32(float* y, const float* x, const int64_t* mask, int64_t mask_period,
                                   const ipdm_sched_t* dev_sched, float scale, int64_t n, void* stream) {
  IPDM_REQUIRE(n >= 0);
  if (n == 0) return IPDM_OK;
  IPDM_REQUIRE(y && x && (!mask || mask_period > 0));
  hipLaunchKernelGGL(axpy_sched_kernel, dim3(ipdm_ew_grid(n, 256)), dim3(256), 0, ipdm_stream(stream), y, x,
                     reinterpret_cast<const long*>(mask), (long long)mask_period, dev_sched, scale, (long long)n);
  return ipdm_launch_status();
}

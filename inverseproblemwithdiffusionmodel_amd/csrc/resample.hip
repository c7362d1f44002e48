// upfirdn2d + fused bias/activation for gfx950.
//
// Both ops are HBM-bound (4*(in+out) bytes per upfirdn2d call, 8 bytes/element for
// bias-act).  upfirdn2d stages one input tile (+halo) per workgroup in LDS with coalesced row loads,
// keeps the flipped, zero-extended FIR taps in LDS, and lets each thread produce several outputs of
// one (plane, tile); only the polyphase taps that hit a non-inserted sample are visited.  Anything
// outside the specialised (up, down, taps<=4) modes takes a generic one-thread-per-output kernel
// with the reference's clamped tap ranges.
//
// Semantics follow op/upfirdn2d.py:168-209 (upfirdn2d_native) / op/upfirdn2d_kernel.cu:49-105 and
// op/fused_bias_act_kernel.cu:19-49 of the reference.
#include "ipdm_common.h"

namespace {

__host__ __device__ __forceinline__ int floor_div_i(int a, int b) {
  int c = a / b;
  if (c * b > a) c--;
  return c;
}

// ---------------------------------------------------------------------------------------------
template <int UP, int DOWN, int K, int TOH, int TOW>
__global__ __launch_bounds__(256) void upfirdn2d_tiled_kernel(
    const float* __restrict__ in, const float* __restrict__ kernel, float* __restrict__ out,
    int in_h, int in_w, int out_h, int out_w, int kernel_h, int kernel_w, int pad_x0, int pad_y0) {
  constexpr int TIH = ((TOH - 1) * DOWN + K - 1) / UP + 2;
  constexpr int TIW = ((TOW - 1) * DOWN + K - 1) / UP + 2;
  constexpr int NT = (K + UP - 1) / UP;     // taps per axis that can hit a real sample
  __shared__ float sk[K][K];
  __shared__ float sx[TIH][TIW + 1];

  const int plane = blockIdx.x;
  const int oy0 = blockIdx.y * TOH;
  const int ox0 = blockIdx.z * TOW;
  const int tid = threadIdx.x;

  // flipped taps, zero-extended to K x K
  if (tid < K * K) {
    int ky = tid / K, kx = tid % K;
    float v = 0.f;
    if (ky < kernel_h && kx < kernel_w) v = kernel[(kernel_h - 1 - ky) * kernel_w + (kernel_w - 1 - kx)];
    sk[ky][kx] = v;
  }
  // first input row / column any output of this tile touches
  const int tin_y0 = floor_div_i(oy0 * DOWN + UP - 1 - pad_y0, UP);
  const int tin_x0 = floor_div_i(ox0 * DOWN + UP - 1 - pad_x0, UP);
  const float* src = in + (size_t)plane * in_h * in_w;
  for (int i = tid; i < TIH * TIW; i += 256) {
    int r = i / TIW, c = i - r * TIW;
    int gy = tin_y0 + r, gx = tin_x0 + c;
    float v = 0.f;
    if (gy >= 0 && gy < in_h && gx >= 0 && gx < in_w) v = src[(size_t)gy * in_w + gx];
    sx[r][c] = v;
  }
  __syncthreads();

  float* dst = out + (size_t)plane * out_h * out_w;
  for (int i = tid; i < TOH * TOW; i += 256) {
    int ty = i / TOW, tx = i - ty * TOW;
    int oy = oy0 + ty, ox = ox0 + tx;
    if (oy >= out_h || ox >= out_w) continue;
    int fy = floor_div_i(oy * DOWN + UP - 1 - pad_y0, UP);     // first contributing input row
    int fx = floor_div_i(ox * DOWN + UP - 1 - pad_x0, UP);
    int jy0 = fy * UP - (oy * DOWN - pad_y0);                  // its tap index in the flipped kernel
    int jx0 = fx * UP - (ox * DOWN - pad_x0);
    int ry = fy - tin_y0, rx = fx - tin_x0;
    float acc = 0.f;
#pragma unroll
    for (int a = 0; a < NT; ++a) {
      int jy = jy0 + a * UP;
      if (K % UP != 0 && jy >= K) break;
#pragma unroll
      for (int b = 0; b < NT; ++b) {
        int jx = jx0 + b * UP;
        if (K % UP != 0 && jx >= K) break;
        acc += sx[ry + a][rx + b] * sk[jy][jx];
      }
    }
    dst[(size_t)oy * out_w + ox] = acc;
  }
}

// ---------------------------------------------------------------------------------------------
struct UpfirdnParams {
  int up_x, up_y, down_x, down_y, pad_x0, pad_y0;
  int major, in_h, in_w, minor, kernel_h, kernel_w, out_h, out_w;
};

__global__ __launch_bounds__(256) void upfirdn2d_generic_kernel(const float* __restrict__ in,
                                                                const float* __restrict__ kernel,
                                                                float* __restrict__ out, UpfirdnParams p) {
  const int64_t total = (int64_t)p.major * p.out_h * p.out_w * p.minor;
  for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total;
       idx += (int64_t)gridDim.x * blockDim.x) {
    int mi = (int)(idx % p.minor);
    int64_t t = idx / p.minor;
    int ox = (int)(t % p.out_w);
    t /= p.out_w;
    int oy = (int)(t % p.out_h);
    int major = (int)(t / p.out_h);

    int mid_y = oy * p.down_y + p.up_y - 1 - p.pad_y0;
    int fy = floor_div_i(mid_y, p.up_y);
    int ky0 = mid_y + p.kernel_h - (fy + 1) * p.up_y;           // unflipped tap index of row fy
    int mid_x = ox * p.down_x + p.up_x - 1 - p.pad_x0;
    int fx = floor_div_i(mid_x, p.up_x);
    int kx0 = mid_x + p.kernel_w - (fx + 1) * p.up_x;

    float acc = 0.f;
    for (int a = 0, ky = ky0; ky >= 0; ++a, ky -= p.up_y) {
      int gy = fy + a;
      if (gy < 0 || ky >= p.kernel_h) continue;
      if (gy >= p.in_h) break;
      for (int b = 0, kx = kx0; kx >= 0; ++b, kx -= p.up_x) {
        int gx = fx + b;
        if (gx < 0 || kx >= p.kernel_w) continue;
        if (gx >= p.in_w) break;
        acc += in[(((size_t)major * p.in_h + gy) * p.in_w + gx) * p.minor + mi] * kernel[ky * p.kernel_w + kx];
      }
    }
    out[idx] = acc;
  }
}

template <int UP, int DOWN, int K, int TOH, int TOW>
int launch_tiled(const float* in, const float* kernel, float* out, const UpfirdnParams& p, hipStream_t s) {
  dim3 grid(p.major, (p.out_h + TOH - 1) / TOH, (p.out_w + TOW - 1) / TOW);
  hipLaunchKernelGGL((upfirdn2d_tiled_kernel<UP, DOWN, K, TOH, TOW>), grid, dim3(256), 0, s, in, kernel, out,
                     p.in_h, p.in_w, p.out_h, p.out_w, p.kernel_h, p.kernel_w, p.pad_x0, p.pad_y0);
  return ipdm_launch_status();
}

// ---------------------------------------------------------------------------------------------
template <bool VEC>
__global__ __launch_bounds__(256) void bias_act_kernel(const float* __restrict__ x, const float* __restrict__ b,
                                                       const float* __restrict__ ref, float* __restrict__ y,
                                                       int64_t n, int step_b, int size_b, int code, float alpha,
                                                       float scale) {
  constexpr int V = VEC ? 4 : 1;
  const int64_t nv = n / V;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < nv; i += (int64_t)gridDim.x * blockDim.x) {
    float v[V], r[V];
    if constexpr (VEC) {
      float4 t = reinterpret_cast<const float4*>(x)[i];
      v[0] = t.x; v[1] = t.y; v[2] = t.z; v[3] = t.w;
      if (ref) {
        float4 q = reinterpret_cast<const float4*>(ref)[i];
        r[0] = q.x; r[1] = q.y; r[2] = q.z; r[3] = q.w;
      }
    } else {
      v[0] = x[i];
      if (ref) r[0] = ref[i];
    }
    if (b) {
      float bb = b[((i * V) / step_b) % size_b];   // VEC requires step_b % 4 == 0: one bias per vector
#pragma unroll
      for (int j = 0; j < V; ++j) v[j] += bb;
    }
#pragma unroll
    for (int j = 0; j < V; ++j) {
      float t = v[j], o;
      switch (code) {
        case 30: o = t > 0.f ? t : t * alpha; break;
        case 31: o = (ref ? r[j] : 0.f) > 0.f ? t : t * alpha; break;
        case 12:
        case 32: o = 0.f; break;
        default: o = t; break;
      }
      v[j] = o * scale;
    }
    if constexpr (VEC) reinterpret_cast<float4*>(y)[i] = make_float4(v[0], v[1], v[2], v[3]);
    else y[i] = v[0];
  }
}

}  // namespace

extern "C" int ipdm_upfirdn2d_f32(const float* in, const float* kernel, float* out, int major, int in_h, int in_w,
                                  int minor, int kernel_h, int kernel_w, int up_x, int up_y, int down_x, int down_y,
                                  int pad_x0, int pad_x1, int pad_y0, int pad_y1, void* stream) {
  IPDM_REQUIRE(in && kernel && out);
  IPDM_REQUIRE(major >= 0 && in_h > 0 && in_w > 0 && minor > 0 && kernel_h > 0 && kernel_w > 0);
  IPDM_REQUIRE(up_x > 0 && up_y > 0 && down_x > 0 && down_y > 0);
  UpfirdnParams p;
  p.up_x = up_x; p.up_y = up_y; p.down_x = down_x; p.down_y = down_y; p.pad_x0 = pad_x0; p.pad_y0 = pad_y0;
  p.major = major; p.in_h = in_h; p.in_w = in_w; p.minor = minor; p.kernel_h = kernel_h; p.kernel_w = kernel_w;
  p.out_h = (in_h * up_y + pad_y0 + pad_y1 - kernel_h) / down_y + 1;
  p.out_w = (in_w * up_x + pad_x0 + pad_x1 - kernel_w) / down_x + 1;
  IPDM_REQUIRE(p.out_h > 0 && p.out_w > 0);
  if (major == 0) return IPDM_OK;
  hipStream_t s = ipdm_stream(stream);
  const bool sq = up_x == up_y && down_x == down_y && minor == 1 && kernel_h <= 4 && kernel_w <= 4;
  if (sq && up_x == 1 && down_x == 2) return launch_tiled<1, 2, 4, 16, 64>(in, kernel, out, p, s);
  if (sq && up_x == 2 && down_x == 1) return launch_tiled<2, 1, 4, 32, 64>(in, kernel, out, p, s);
  if (sq && up_x == 1 && down_x == 1) return launch_tiled<1, 1, 4, 32, 64>(in, kernel, out, p, s);
  const int64_t total = (int64_t)major * p.out_h * p.out_w * minor;
  hipLaunchKernelGGL(upfirdn2d_generic_kernel, dim3(ipdm_ew_grid(total, 256)), dim3(256), 0, s, in, kernel, out, p);
  return ipdm_launch_status();
}

extern "C" int ipdm_fused_bias_act_f32(const float* x, const float* b, const float* ref, float* y, int64_t n,
                                       int step_b, int size_b, int act, int grad, float alpha, float scale,
                                       void* stream) {
  IPDM_REQUIRE(n >= 0);
  if (n == 0) return IPDM_OK;
  IPDM_REQUIRE(x && y);
  if (size_b <= 0) b = nullptr;
  if (b) IPDM_REQUIRE(step_b > 0);
  const int code = act * 10 + grad;
  hipStream_t s = ipdm_stream(stream);
  const bool aligned = ((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(y) |
                         reinterpret_cast<uintptr_t>(ref)) & 15) == 0;
  const bool vec = aligned && (n % 4 == 0) && (!b || step_b % 4 == 0);
  if (vec)
    hipLaunchKernelGGL(bias_act_kernel<true>, dim3(ipdm_ew_grid(n / 4, 256)), dim3(256), 0, s, x, b, ref, y, n, step_b,
                       size_b, code, alpha, scale);
  else
    hipLaunchKernelGGL(bias_act_kernel<false>, dim3(ipdm_ew_grid(n, 256)), dim3(256), 0, s, x, b, ref, y, n, step_b,
                       size_b, code, alpha, scale);
  return ipdm_launch_status();
}

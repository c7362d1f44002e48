// 3-D helpers of the temporal prior (NCSN3DShallow, 8 x 8 x T volumes): 5x5x5 max-pool and the tap gather that
// turns the strided / transposed temporal convolutions into 1x1 convolutions for the MFMA kernel.  HBM-bound.
//
// Reference semantics: nn.MaxPool3d(kernel_size=5, stride=1, padding=2) (CRPBlock of ncsn/models/layers3d.py);
// nn.Conv3d(k=(1,1,4), stride=(1,1,2), padding=(0,0,1)) and nn.ConvTranspose3d(same) (ncsn/models/ncsn3d.py:176-177).
#include "ipdm_common.h"

namespace {

// one workgroup per (b, c) volume, separable max over the three axes through two LDS images
__global__ __launch_bounds__(256) void maxpool3d5_kernel(const float* __restrict__ x, float* __restrict__ y, int D, int H,
                                                         int W) {
  extern __shared__ float sm[];
  const int n = D * H * W;
  float* a = sm;
  float* b = sm + n;
  const float* p = x + (size_t)blockIdx.x * n;
  for (int i = threadIdx.x; i < n; i += 256) a[i] = p[i];
  __syncthreads();
  for (int i = threadIdx.x; i < n; i += 256) {           // along W
    const int w = i % W, base = i - w;
    float m = a[i];
    for (int o = -2; o <= 2; ++o) {
      const int ww = w + o;
      if (ww >= 0 && ww < W) m = fmaxf(m, a[base + ww]);
    }
    b[i] = m;
  }
  __syncthreads();
  for (int i = threadIdx.x; i < n; i += 256) {           // along H
    const int h = (i / W) % H;
    float m = b[i];
    for (int o = -2; o <= 2; ++o) {
      const int hh = h + o;
      if (hh >= 0 && hh < H) m = fmaxf(m, b[i + o * W]);
    }
    a[i] = m;
  }
  __syncthreads();
  float* q = y + (size_t)blockIdx.x * n;
  for (int i = threadIdx.x; i < n; i += 256) {           // along D
    const int d = i / (H * W);
    float m = a[i];
    for (int o = -2; o <= 2; ++o) {
      const int dd = d + o;
      if (dd >= 0 && dd < D) m = fmaxf(m, a[i + o * H * W]);
    }
    q[i] = m;
  }
}

// x [planes][S][T_in] -> out [planes][4][S][T_out]
//   mode 0 (stride-2 conv, pad 1):        out[p][k][s][t] = x[p][s][2t - 1 + k]          T_out = T_in / 2
//   mode 1 (stride-2 transposed, pad 1):  out[p][k][s][t] = xup[t + 1 - k], xup[u] = x[p][s][u/2] for even u
__global__ __launch_bounds__(256) void temporal_taps_kernel(const float* __restrict__ x, float* __restrict__ out,
                                                            int64_t total, int S, int T_in, int T_out, int mode) {
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    const int t = (int)(i % T_out);
    int64_t r = i / T_out;
    const int s = (int)(r % S);
    r /= S;
    const int k = (int)(r % 4);
    const int64_t p = r / 4;
    float v = 0.f;
    if (mode == 0) {
      const int ti = 2 * t - 1 + k;
      if (ti >= 0 && ti < T_in) v = x[(p * S + s) * T_in + ti];
    } else {
      const int u = t + 1 - k;
      if (u >= 0 && (u & 1) == 0 && (u >> 1) < T_in) v = x[(p * S + s) * T_in + (u >> 1)];
    }
    out[i] = v;
  }
}

// trilinear resize, align_corners=True (F.interpolate(mode='trilinear') of the 3-D MSF block, layers3d.py:185,214): one
// output element per thread, the eight taps combined in ATen's order (width, then height, then depth)
__global__ __launch_bounds__(256) void trilinear_kernel(const float* __restrict__ x, float* out, long long total, int id,
                                                        int ih, int iw, int od, int oh, int ow, float sd, float sh, float sw,
                                                        int accumulate, int act, float* amax) {
  // (grid.y = images when amax is wanted; total = outputs per grid.y slice)
  float amx = 0.f;
  const long long base = (long long)blockIdx.y * total;
  for (long long k = (long long)blockIdx.x * 256 + threadIdx.x; k < total; k += (long long)gridDim.x * 256) {
    const long long i = base + k;
    const int ox = (int)(i % ow);
    long long t = i / ow;
    const int oy = (int)(t % oh);
    t /= oh;
    const int oz = (int)(t % od);
    const long long plane = t / od;
    const float fz = sd * oz, fy = sh * oy, fx = sw * ox;
    const int z0 = (int)fz, y0 = (int)fy, x0 = (int)fx;
    const int zp = z0 < id - 1 ? 1 : 0, yp = y0 < ih - 1 ? 1 : 0, xp = x0 < iw - 1 ? 1 : 0;
    const float lz1 = fz - z0, lz0 = 1.f - lz1, ly1 = fy - y0, ly0 = 1.f - ly1, lx1 = fx - x0, lx0 = 1.f - lx1;
    const float* p00 = x + ((plane * id + z0) * ih + y0) * (long long)iw + x0;
    const float* p01 = p00 + (long long)yp * iw;
    const float* p10 = p00 + (long long)zp * ih * iw;
    const float* p11 = p10 + (long long)yp * iw;
    float v = lz0 * (ly0 * (lx0 * p00[0] + lx1 * p00[xp]) + ly1 * (lx0 * p01[0] + lx1 * p01[xp])) +
              lz1 * (ly0 * (lx0 * p10[0] + lx1 * p10[xp]) + ly1 * (lx0 * p11[0] + lx1 * p11[xp]));
    if (accumulate) v += out[i];
    v = ipdm_act(v, act);
    amx = fmaxf(amx, fabsf(v));
    out[i] = v;
  }
  __shared__ float red[4];
  if (amax) ipdm_amax_commit_block(amx, amax + (size_t)blockIdx.y * IPDM_AMAX_SLOT, (int)blockIdx.x, red);
}

}  // namespace

extern "C" int ipdm_maxpool3d5_f32(const float* x, float* y, int planes, int D, int H, int W, void* stream) {
  IPDM_REQUIRE(planes >= 0 && D > 0 && H > 0 && W > 0);
  if (planes == 0) return IPDM_OK;
  IPDM_REQUIRE(x && y && x != y);
  const size_t lds = (size_t)2 * D * H * W * sizeof(float);
  if (lds > 64 * 1024) return IPDM_EUNSUPPORTED;
  hipLaunchKernelGGL(maxpool3d5_kernel, dim3(planes), dim3(256), lds, ipdm_stream(stream), x, y, D, H, W);
  return ipdm_launch_status();
}

extern "C" int ipdm_temporal_taps_f32(const float* x, float* out, int planes, int S, int T_in, int T_out, int mode,
                                      void* stream) {
  IPDM_REQUIRE(planes >= 0 && S > 0 && T_in > 0 && T_out > 0 && (mode == 0 || mode == 1));
  if (planes == 0) return IPDM_OK;
  IPDM_REQUIRE(x && out && x != out);
  const int64_t total = (int64_t)planes * 4 * S * T_out;
  hipLaunchKernelGGL(temporal_taps_kernel, dim3(ipdm_ew_grid(total, 256)), dim3(256), 0, ipdm_stream(stream), x, out,
                     (long long)total, S, T_in, T_out, mode);
  return ipdm_launch_status();
}

extern "C" int ipdm_trilinear_f32(const float* x, float* out, int planes, int in_d, int in_h, int in_w, int out_d, int out_h,
                                  int out_w, int accumulate, int act, int planes_per_image, float* amax_out, void* stream) {
  IPDM_REQUIRE(planes >= 0 && in_d > 0 && in_h > 0 && in_w > 0 && out_d > 0 && out_h > 0 && out_w > 0);
  if (planes == 0) return IPDM_OK;
  IPDM_REQUIRE(x && out && x != out);
  IPDM_REQUIRE(!amax_out || (planes_per_image > 0 && planes % planes_per_image == 0 && planes / planes_per_image <= 65535));
  const int n_img = amax_out ? planes / planes_per_image : 1;
  const float sd = out_d > 1 ? (float)(in_d - 1) / (float)(out_d - 1) : 0.f;
  const float sh = out_h > 1 ? (float)(in_h - 1) / (float)(out_h - 1) : 0.f;
  const float sw = out_w > 1 ? (float)(in_w - 1) / (float)(out_w - 1) : 0.f;
  const int64_t total = (int64_t)planes * out_d * out_h * out_w;
  const int64_t per = total / n_img;
  int64_t gx = (per + 255) / 256, cap = (2048 + n_img - 1) / n_img;
  gx = gx < 1 ? 1 : (gx > cap ? cap : gx);
  hipLaunchKernelGGL(trilinear_kernel, dim3((unsigned)gx, (unsigned)n_img), dim3(256), 0, ipdm_stream(stream), x, out,
                     (long long)per, in_d, in_h, in_w, out_d, out_h, out_w, sd, sh, sw, accumulate, act, amax_out);
  return ipdm_launch_status();
}

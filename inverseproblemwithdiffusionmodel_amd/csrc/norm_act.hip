// Score-network glue kernels for gfx950: InstanceNorm++ statistics -> per-(image, channel) affine
// coefficients, fused affine+activation, element-wise helpers, 5x5 max-pool, 2x2 mean-pool and
// align_corners bilinear resize(+accumulate).  All HBM/L2-bound, NCHW planar float32.
//
// Reference semantics: ncsn/models/normalization.py:150-176 (InstanceNorm2dPlus),
// ncsn/models/layers.py:11-23 (activations), :62-83 (CRPBlock MaxPool2d(5,1,2)), :165-184 (MSFBlock
// F.interpolate bilinear align_corners=True), :309-313 (ConvMeanPool mean), ncsnv2.py:270-271,295-297.
#include "ipdm_common.h"

namespace {

// ---- InstanceNorm++ ---------------------------------------------------------------------------
// pass 1: one workgroup per (b, c) plane: mean and 1/sqrt(biased var + 1e-5), two sweeps.  Planes of up to 128x128
// are held in registers between the sweeps (16 float4 per thread): with thousands of 64 KiB planes in flight the
// second sweep would otherwise miss L2 and read HBM again.  Same arithmetic and summation order on both paths.
template <bool REG>
__global__ __launch_bounds__(256) void plane_stats_kernel(const float* __restrict__ x, float* __restrict__ coef,
                                                          int HW) {
  __shared__ double red[4];
  const float* p = x + (size_t)blockIdx.x * HW;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const bool vec = REG || ((HW % 4 == 0) && ((reinterpret_cast<uintptr_t>(p) & 15) == 0));
  float4 keep[16];
  float s = 0.f;
  if constexpr (REG) {
    const int n4 = HW / 4;
#pragma unroll
    for (int k = 0; k < 16; ++k) {
      const int i = tid + k * 256;
      keep[k] = i < n4 ? reinterpret_cast<const float4*>(p)[i] : make_float4(0.f, 0.f, 0.f, 0.f);
    }
#pragma unroll
    for (int k = 0; k < 16; ++k)
      if (tid + k * 256 < n4) s += (keep[k].x + keep[k].y) + (keep[k].z + keep[k].w);
  } else if (vec) {
    for (int i = tid; i < HW / 4; i += 256) {
      float4 v = reinterpret_cast<const float4*>(p)[i];
      s += (v.x + v.y) + (v.z + v.w);
    }
  } else {
    for (int i = tid; i < HW; i += 256) s += p[i];
  }
  double ds = ipdm_wave_sum((double)s);
  if (lane == 0) red[wave] = ds;
  __syncthreads();
  const float mean = (float)((red[0] + red[1] + red[2] + red[3]) / (double)HW);
  __syncthreads();
  float q = 0.f;
  if constexpr (REG) {
    const int n4 = HW / 4;
#pragma unroll
    for (int k = 0; k < 16; ++k) {
      if (tid + k * 256 < n4) {
        const float a = keep[k].x - mean, b = keep[k].y - mean, c = keep[k].z - mean, d = keep[k].w - mean;
        q += (a * a + b * b) + (c * c + d * d);
      }
    }
  } else if (vec) {
    for (int i = tid; i < HW / 4; i += 256) {
      float4 v = reinterpret_cast<const float4*>(p)[i];
      float a = v.x - mean, b = v.y - mean, c = v.z - mean, d = v.w - mean;
      q += (a * a + b * b) + (c * c + d * d);
    }
  } else {
    for (int i = tid; i < HW; i += 256) {
      float a = p[i] - mean;
      q += a * a;
    }
  }
  double dq = ipdm_wave_sum((double)q);
  if (lane == 0) red[wave] = dq;
  __syncthreads();
  if (tid == 0) {
    float var = (float)((red[0] + red[1] + red[2] + red[3]) / (double)HW);
    coef[(size_t)blockIdx.x * 3 + 0] = mean;
    coef[(size_t)blockIdx.x * 3 + 1] = 1.0f / sqrtf(var + 1e-5f);
  }
}

// small planes (<= 1024 elements: the 32- and 16-pixel stages): ONE WAVE per plane, the plane in registers (up to four float4
// per lane), both sweeps reduced by wave butterflies -- no LDS, no barrier.  A 256-thread workgroup per 256-element plane spent
// its time in two block reductions (28 us per call for 14 MB); same two-sweep arithmetic, float64 butterflies.
__global__ __launch_bounds__(256) void plane_stats_small_kernel(const float* __restrict__ x, float* __restrict__ coef,
                                                                int planes, int HW) {
  const int lane = threadIdx.x & 63;
  const int pl = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (pl >= planes) return;
  const float4* p = reinterpret_cast<const float4*>(x + (size_t)pl * HW);
  const int n4 = HW / 4;
  float4 keep[4];
  float s = 0.f;
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const int i = lane + k * 64;
    keep[k] = i < n4 ? p[i] : make_float4(0.f, 0.f, 0.f, 0.f);
    if (i < n4) s += (keep[k].x + keep[k].y) + (keep[k].z + keep[k].w);
  }
  const float mean = (float)(ipdm_wave_sum((double)s) / (double)HW);
  float q = 0.f;
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    if (lane + k * 64 < n4) {
      const float a = keep[k].x - mean, b = keep[k].y - mean, c = keep[k].z - mean, d = keep[k].w - mean;
      q += (a * a + b * b) + (c * c + d * d);
    }
  }
  const float var = (float)(ipdm_wave_sum((double)q) / (double)HW);
  if (lane == 0) {
    coef[(size_t)pl * 3 + 0] = mean;
    coef[(size_t)pl * 3 + 1] = 1.0f / sqrtf(var + 1e-5f);
  }
}

// pass 2: one workgroup per image: cross-channel mean / unbiased variance of the plane means, then
// coef[b][c] = (mu, gamma*rstd, beta + gamma*alpha*(mu - m)/sqrt(v + 1e-5))
// amax_bound (optional, a maxima vector [B][IPDM_AMAX_SLOT]): an upper bound of max |normalised value| over the image, from the coefficients alone --
// |x - mu| * rstd < sqrt(HW) for every element of a plane (n - 1 squared deviations cannot exceed n times the variance), so
// |(x - mu) * scale + shift| < |scale / rstd| * sqrt(HW) + |shift|, and every activation code shrinks |.|.  What the consuming
// f16x2 convolution takes as in_amax (conv_kernel.h, hx_dynamic_scale): a bound is enough there, and this one costs nothing.
__global__ __launch_bounds__(256) void instnorm_plus_coef_kernel(float* __restrict__ coef,
                                                                 const float* __restrict__ alpha,
                                                                 const float* __restrict__ gamma,
                                                                 const float* __restrict__ beta, int C,
                                                                 float* __restrict__ amax_bound, float sqrt_hw) {
  __shared__ double red[4];
  __shared__ float redf[4];
  float* cb = coef + (size_t)blockIdx.x * C * 3;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  double s = 0.0;
  for (int c = tid; c < C; c += 256) s += (double)cb[c * 3];
  s = ipdm_wave_sum(s);
  if (lane == 0) red[wave] = s;
  __syncthreads();
  const float m = (float)((red[0] + red[1] + red[2] + red[3]) / (double)C);
  __syncthreads();
  double q = 0.0;
  for (int c = tid; c < C; c += 256) {
    double d = (double)cb[c * 3] - (double)m;
    q += d * d;
  }
  q = ipdm_wave_sum(q);
  if (lane == 0) red[wave] = q;
  __syncthreads();
  const float v = (float)((red[0] + red[1] + red[2] + red[3]) / (double)(C - 1));   // C == 1 -> NaN, as torch.var
  const float inv = 1.0f / sqrtf(v + 1e-5f);
  float bound = 0.f;
  for (int c = tid; c < C; c += 256) {
    float mu = cb[c * 3], rstd = cb[c * 3 + 1];
    float g = gamma[c];
    float mn = (mu - m) * inv;
    const float shift = (beta ? beta[c] : 0.f) + g * (mn * alpha[c]);
    cb[c * 3 + 1] = g * rstd;
    cb[c * 3 + 2] = shift;
    bound = fmaxf(bound, fabsf(g) * sqrt_hw + fabsf(shift));
  }
  if (amax_bound) {                                            // (uniform)
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) bound = fmaxf(bound, __shfl_xor(bound, o, 64));
    if (lane == 0) redf[wave] = bound;
    __syncthreads();
    if (tid < IPDM_AMAX_WAYS)                                    // every way of the image's slot: no zeroing needed
      amax_bound[(size_t)blockIdx.x * IPDM_AMAX_SLOT + tid * IPDM_AMAX_WAY_STRIDE] = fmaxf(fmaxf(redf[0], redf[1]), fmaxf(redf[2], redf[3]));
  }
}

// pass 1 from the producing convolution's epilogue partials ([B][C][P][3] = count, mean, sum of squared deviations per tile
// group): per plane, the partials are merged in index order with the pooled-variance update (exact in real arithmetic,
// float64 here), giving the same (mean, 1/sqrt(biased var + 1e-5)) plane_stats_kernel writes -- without reading the tensor
__global__ __launch_bounds__(256) void plane_stats_from_partials_kernel(const float* __restrict__ part, float* __restrict__ coef,
                                                                        int planes, int P) {
  // one wave per plane: lane l merges partials l, l+64, ... in order, then a butterfly merges the 64 lanes (the pairwise
  // update is symmetric, so every lane ends with the same triple): a fixed tree, whatever the batch
  const int lane = threadIdx.x & 63;
  const int pl = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (pl >= planes) return;
  const float* pp = part + (size_t)pl * P * 3;
  double n = 0.0, mean = 0.0, m2 = 0.0;
  for (int i = lane; i < P; i += 64) {
    const double nb = pp[i * 3], mb = pp[i * 3 + 1], qb = pp[i * 3 + 2];
    if (nb > 0.0) {
      const double nn = n + nb, d = mb - mean;
      mean += d * (nb / nn);
      m2 += qb + d * d * (n * nb / nn);
      n = nn;
    }
  }
#pragma unroll
  for (int m = 1; m < 64; m <<= 1) {
    const double nb = __shfl_xor(n, m, 64), mb = __shfl_xor(mean, m, 64), qb = __shfl_xor(m2, m, 64);
    const double nn = n + nb;
    if (nn > 0.0) {
      // symmetric form: both partners compute the same numbers
      const double d = mb - mean;
      const double new_mean = (n * mean + nb * mb) / nn;
      m2 = (m2 + qb) + d * d * (n * nb / nn);
      mean = new_mean;
      n = nn;
    }
  }
  if (lane == 0) {
    coef[(size_t)pl * 3 + 0] = (float)mean;
    coef[(size_t)pl * 3 + 1] = 1.0f / sqrtf((float)(m2 / n) + 1e-5f);
  }
}

__global__ __launch_bounds__(256) void affine_act_kernel(const float* x, const float* __restrict__ coef, float* y,
                                                         int HW, int act) {
  // grid.x = plane tiles, grid.y = planes
  const size_t plane = blockIdx.y;
  const float mu = coef[plane * 3], sc = coef[plane * 3 + 1], sh = coef[plane * 3 + 2];
  const float* p = x + plane * HW;
  float* o = y + plane * HW;
  if ((HW & 3) == 0) {
    for (int i = blockIdx.x * 256 + threadIdx.x; i < HW / 4; i += gridDim.x * 256) {
      float4 v = reinterpret_cast<const float4*>(p)[i];
      v.x = ipdm_act((v.x - mu) * sc + sh, act);
      v.y = ipdm_act((v.y - mu) * sc + sh, act);
      v.z = ipdm_act((v.z - mu) * sc + sh, act);
      v.w = ipdm_act((v.w - mu) * sc + sh, act);
      reinterpret_cast<float4*>(o)[i] = v;
    }
  } else {
    for (int i = blockIdx.x * 256 + threadIdx.x; i < HW; i += gridDim.x * 256) o[i] = ipdm_act((p[i] - mu) * sc + sh, act);
  }
}

// affine + activation of torch.cat([x1, x2], dim=1) without the concatenation: plane (b, c) of the output reads channel c of x1
// (c < C1) or channel c - C1 of x2; the output IS the concatenated, normalised tensor (written once)
__global__ __launch_bounds__(256) void affine_act_cat_kernel(const float* x1, int C1, const float* x2, int C2,
                                                             const float* __restrict__ coef, float* y, int HW, int act) {
  const size_t plane = blockIdx.y;
  const int Ct = C1 + C2;
  const size_t b = plane / Ct;
  const int c = (int)(plane % Ct);
  const float mu = coef[plane * 3], sc = coef[plane * 3 + 1], sh = coef[plane * 3 + 2];
  const float* p = c < C1 ? x1 + (b * C1 + c) * HW : x2 + (b * C2 + (c - C1)) * HW;
  float* o = y + plane * HW;
  if ((HW & 3) == 0) {
    for (int i = blockIdx.x * 256 + threadIdx.x; i < HW / 4; i += gridDim.x * 256) {
      float4 v = reinterpret_cast<const float4*>(p)[i];
      v.x = ipdm_act((v.x - mu) * sc + sh, act);
      v.y = ipdm_act((v.y - mu) * sc + sh, act);
      v.z = ipdm_act((v.z - mu) * sc + sh, act);
      v.w = ipdm_act((v.w - mu) * sc + sh, act);
      reinterpret_cast<float4*>(o)[i] = v;
    }
  } else {
    for (int i = blockIdx.x * 256 + threadIdx.x; i < HW; i += gridDim.x * 256) o[i] = ipdm_act((p[i] - mu) * sc + sh, act);
  }
}

// ---- generic element-wise -------------------------------------------------------------------
template <typename F>
__global__ __launch_bounds__(256) void ew1_kernel(const float* x, float* y, int64_t n, F f) {
  const int64_t nv = n / 4;
  const bool vec = ((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(y)) & 15) == 0;
  const int64_t tid = (int64_t)blockIdx.x * 256 + threadIdx.x, stride = (int64_t)gridDim.x * 256;
  if (vec) {
    for (int64_t i = tid; i < nv; i += stride) {
      float4 v = reinterpret_cast<const float4*>(x)[i];
      v.x = f(v.x); v.y = f(v.y); v.z = f(v.z); v.w = f(v.w);
      reinterpret_cast<float4*>(y)[i] = v;
    }
    for (int64_t i = nv * 4 + tid; i < n; i += stride) y[i] = f(x[i]);
  } else {
    for (int64_t i = tid; i < n; i += stride) y[i] = f(x[i]);
  }
}

__global__ __launch_bounds__(256) void add_kernel(const float* x, const float* y, float* out, int64_t n) {
  const int64_t nv = n / 4;
  const bool vec = ((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(y) | reinterpret_cast<uintptr_t>(out)) & 15) == 0;
  const int64_t tid = (int64_t)blockIdx.x * 256 + threadIdx.x, stride = (int64_t)gridDim.x * 256;
  if (vec) {
    for (int64_t i = tid; i < nv; i += stride) {
      float4 a = reinterpret_cast<const float4*>(x)[i], b = reinterpret_cast<const float4*>(y)[i];
      reinterpret_cast<float4*>(out)[i] = make_float4(a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w);
    }
    for (int64_t i = nv * 4 + tid; i < n; i += stride) out[i] = x[i] + y[i];
  } else {
    for (int64_t i = tid; i < n; i += stride) out[i] = x[i] + y[i];
  }
}

__global__ __launch_bounds__(256) void div_sigma_kernel(const float* x, const float* __restrict__ sigmas,
                                                        const int64_t* __restrict__ labels, float* out,
                                                        int64_t sample_elems) {
  const float sg = labels ? sigmas[labels[blockIdx.y]] : sigmas[blockIdx.y];
  const float* p = x + (size_t)blockIdx.y * sample_elems;
  float* o = out + (size_t)blockIdx.y * sample_elems;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < sample_elems; i += (int64_t)gridDim.x * 256) o[i] = p[i] / sg;
}

// ---- pooling / resize -------------------------------------------------------------------------
constexpr int MP_T = 32;
__global__ __launch_bounds__(256) void maxpool5_kernel(const float* __restrict__ x, float* __restrict__ y, int H, int W) {
  __shared__ float tile[MP_T + 4][MP_T + 4 + 1];
  __shared__ float rowmax[MP_T + 4][MP_T + 1];
  const int tiles_x = (W + MP_T - 1) / MP_T;
  const int ty0 = (blockIdx.x / tiles_x) * MP_T, tx0 = (blockIdx.x % tiles_x) * MP_T;
  const float* p = x + (size_t)blockIdx.y * H * W;
  float* o = y + (size_t)blockIdx.y * H * W;
  for (int i = threadIdx.x; i < (MP_T + 4) * (MP_T + 4); i += 256) {
    int r = i / (MP_T + 4), c = i % (MP_T + 4);
    int gy = ty0 + r - 2, gx = tx0 + c - 2;
    tile[r][c] = (gy >= 0 && gy < H && gx >= 0 && gx < W) ? p[(size_t)gy * W + gx] : -INFINITY;
  }
  __syncthreads();
  for (int i = threadIdx.x; i < (MP_T + 4) * MP_T; i += 256) {
    int r = i / MP_T, c = i % MP_T;
    float m = fmaxf(fmaxf(tile[r][c], tile[r][c + 1]), fmaxf(tile[r][c + 2], tile[r][c + 3]));
    rowmax[r][c] = fmaxf(m, tile[r][c + 4]);
  }
  __syncthreads();
  for (int i = threadIdx.x; i < MP_T * MP_T; i += 256) {
    int r = i / MP_T, c = i % MP_T;
    int gy = ty0 + r, gx = tx0 + c;
    if (gy < H && gx < W) {
      float m = fmaxf(fmaxf(rowmax[r][c], rowmax[r + 1][c]), fmaxf(rowmax[r + 2][c], rowmax[r + 3][c]));
      o[(size_t)gy * W + gx] = fmaxf(m, rowmax[r + 4][c]);
    }
  }
}

// float4 path (W % 4 == 0): a thread owns 4 adjacent columns x MP_R rows and slides a five-row window of row maxima
// down its strip in registers -- no LDS, no barriers; the three float4 loads per row overlap with the neighbouring
// threads' and are served by L1.  max is exact, so the result is bit-identical to the tiled kernel.
constexpr int MP_R = 8;
__global__ __launch_bounds__(256) void maxpool5_strip_kernel(const float* __restrict__ x, float* __restrict__ y, int H,
                                                             int W, int64_t n_items) {
  const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (idx >= n_items) return;
  const int WG = W / 4, HS = (H + MP_R - 1) / MP_R;
  const int xg = (int)(idx % WG);
  const int64_t t = idx / WG;
  const int strip = (int)(t % HS);
  const int64_t plane = t / HS;
  const int x0 = xg * 4, y0 = strip * MP_R;
  const float* p = x + plane * (int64_t)H * W;
  float* o = y + plane * (int64_t)H * W;
  const float NI = -INFINITY;
  float rm[5][4];
#pragma unroll
  for (int k = 0; k < 5; ++k)
#pragma unroll
    for (int j = 0; j < 4; ++j) rm[k][j] = NI;
#pragma unroll
  for (int i = 0; i < MP_R + 4; ++i) {
    const int r = y0 - 2 + i;
    float4 a = make_float4(NI, NI, NI, NI), b = a, c = a;
    if (r >= 0 && r < H) {
      const float* row = p + (int64_t)r * W + x0;
      b = *reinterpret_cast<const float4*>(row);
      if (x0 >= 4) a = *reinterpret_cast<const float4*>(row - 4);
      if (x0 + 4 < W) c = *reinterpret_cast<const float4*>(row + 4);
    }
    // v0..v7 = columns x0-2 .. x0+5
    const float v0 = a.z, v1 = a.w, v2 = b.x, v3 = b.y, v4 = b.z, v5 = b.w, v6 = c.x, v7 = c.y;
    const float m12 = fmaxf(v1, v2), m34 = fmaxf(v3, v4), m56 = fmaxf(v5, v6);
    float* cur = rm[i % 5];
    cur[0] = fmaxf(fmaxf(v0, m12), m34);
    cur[1] = fmaxf(fmaxf(m12, m34), v5);
    cur[2] = fmaxf(fmaxf(v2, m34), m56);
    cur[3] = fmaxf(fmaxf(m34, m56), v7);
    if (i >= 4) {                                             // rows r-4 .. r are in the ring: output row r-2
      const int ro = r - 2;
      if (ro < H) {
        float4 out;
        out.x = fmaxf(fmaxf(fmaxf(rm[0][0], rm[1][0]), fmaxf(rm[2][0], rm[3][0])), rm[4][0]);
        out.y = fmaxf(fmaxf(fmaxf(rm[0][1], rm[1][1]), fmaxf(rm[2][1], rm[3][1])), rm[4][1]);
        out.z = fmaxf(fmaxf(fmaxf(rm[0][2], rm[1][2]), fmaxf(rm[2][2], rm[3][2])), rm[4][2]);
        out.w = fmaxf(fmaxf(fmaxf(rm[0][3], rm[1][3]), fmaxf(rm[2][3], rm[3][3])), rm[4][3]);
        *reinterpret_cast<float4*>(o + (int64_t)ro * W + x0) = out;
      }
    }
  }
}

__global__ __launch_bounds__(256) void meanpool2_kernel(const float* __restrict__ x, float* __restrict__ y, int64_t n_out,
                                                        int H, int W) {
  const int OH = H / 2, OW = W / 2;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n_out; i += (int64_t)gridDim.x * 256) {
    int ox = (int)(i % OW);
    int64_t t = i / OW;
    int oy = (int)(t % OH);
    int64_t plane = t / OH;
    const float* p = x + (plane * H + 2 * oy) * (int64_t)W + 2 * ox;
    float2 top = *reinterpret_cast<const float2*>(p);
    float2 bot = *reinterpret_cast<const float2*>(p + W);
    y[i] = (((top.x + bot.x) + top.y) + bot.y) / 4.0f;     // the reference's summation order
  }
}

// max |written value| of a thread -> one atomic max per wave on the image's slot of a maxima vector (ipdm_common.h)
__device__ __forceinline__ void amax_commit(float m, float* image_slot, int way) {
  __shared__ float red[4];
  ipdm_amax_commit_block(m, image_slot, way, red);
}

// (grid.y = images when amax is wanted: a workgroup then stays inside one image; n_out = outputs per grid.y slice)
__global__ __launch_bounds__(256) void bilinear_kernel(const float* __restrict__ x, float* out, int64_t n_out, int ih,
                                                       int iw, int oh, int ow, float sh, float sw, int accumulate, int act,
                                                       float* amax) {
  float amx = 0.f;
  const int64_t base = (int64_t)blockIdx.y * n_out;
  for (int64_t k = (int64_t)blockIdx.x * 256 + threadIdx.x; k < n_out; k += (int64_t)gridDim.x * 256) {
    const int64_t i = base + k;
    int ox = (int)(i % ow);
    int64_t t = i / ow;
    int oy = (int)(t % oh);
    int64_t plane = t / oh;
    float fy = sh * oy, fx = sw * ox;
    int y0 = (int)fy, x0 = (int)fx;
    int yp = y0 < ih - 1 ? 1 : 0, xp = x0 < iw - 1 ? 1 : 0;
    float ly1 = fy - y0, lx1 = fx - x0;
    float ly0 = 1.f - ly1, lx0 = 1.f - lx1;
    const float* p = x + plane * (int64_t)ih * iw;
    float v = ly0 * (lx0 * p[y0 * iw + x0] + lx1 * p[y0 * iw + x0 + xp]) +
              ly1 * (lx0 * p[(y0 + yp) * iw + x0] + lx1 * p[(y0 + yp) * iw + x0 + xp]);
    const float r = ipdm_act(accumulate ? out[i] + v : v, act);
    amx = fmaxf(amx, fabsf(r));
    out[i] = r;
  }
  if (amax) amax_commit(amx, amax + (size_t)blockIdx.y * IPDM_AMAX_SLOT, (int)blockIdx.x);
}

// ow % 4 == 0: four adjacent outputs per thread (one row / plane decomposition, float4 read-modify-write of `out`);
// per-element arithmetic identical to bilinear_kernel
__global__ __launch_bounds__(256) void bilinear4_kernel(const float* __restrict__ x, float* out, int64_t n_items, int ih,
                                                        int iw, int oh, int ow, float sh, float sw, int accumulate, int act,
                                                        float* amax) {
  const int og = ow / 4;
  float amx = 0.f;
  const int64_t base = (int64_t)blockIdx.y * n_items;
  for (int64_t k = (int64_t)blockIdx.x * 256 + threadIdx.x; k < n_items; k += (int64_t)gridDim.x * 256) {
    const int64_t it = base + k;
    const int xg = (int)(it % og);
    const int64_t t = it / og;
    const int oy = (int)(t % oh);
    const int64_t plane = t / oh;
    const float fy = sh * oy;
    const int y0 = (int)fy;
    const int yp = y0 < ih - 1 ? 1 : 0;
    const float ly1 = fy - y0, ly0 = 1.f - ly1;
    const float* r0 = x + (plane * ih + y0) * (int64_t)iw;
    const float* r1 = r0 + yp * iw;
    float* po = out + (plane * oh + oy) * (int64_t)ow + xg * 4;
    float4 acc = accumulate ? *reinterpret_cast<const float4*>(po) : make_float4(0.f, 0.f, 0.f, 0.f);
    float res[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int ox = xg * 4 + j;
      const float fx = sw * ox;
      const int x0 = (int)fx;
      const int xp = x0 < iw - 1 ? 1 : 0;
      const float lx1 = fx - x0, lx0 = 1.f - lx1;
      res[j] = ly0 * (lx0 * r0[x0] + lx1 * r0[x0 + xp]) + ly1 * (lx0 * r1[x0] + lx1 * r1[x0 + xp]);
    }
    float4 o;
    o.x = ipdm_act(accumulate ? acc.x + res[0] : res[0], act);
    o.y = ipdm_act(accumulate ? acc.y + res[1] : res[1], act);
    o.z = ipdm_act(accumulate ? acc.z + res[2] : res[2], act);
    o.w = ipdm_act(accumulate ? acc.w + res[3] : res[3], act);
    amx = fmaxf(fmaxf(amx, fmaxf(fabsf(o.x), fabsf(o.y))), fmaxf(fabsf(o.z), fabsf(o.w)));
    *reinterpret_cast<float4*>(po) = o;
  }
  if (amax) amax_commit(amx, amax + (size_t)blockIdx.y * IPDM_AMAX_SLOT, (int)blockIdx.x);
}

struct ActOp {
  int act;
  __device__ float operator()(float v) const { return ipdm_act(v, act); }
};
struct ScaleShiftOp {
  float a, b;
  __device__ float operator()(float v) const { return a * v + b; }
};

}  // namespace

extern "C" int ipdm_instnorm_plus_coef_f32(const float* x, const float* alpha, const float* gamma, const float* beta,
                                           float* coef, int B, int C, int HW, float* amax_bound, void* stream) {
  IPDM_REQUIRE(B >= 0 && C > 0 && HW > 0);
  if (B == 0) return IPDM_OK;
  IPDM_REQUIRE(x && alpha && gamma && coef);
  hipStream_t s = ipdm_stream(stream);
  // register-resident planes: float4-aligned (HW % 4 == 0 keeps every plane of a 16-byte aligned tensor aligned)
  if (HW % 4 == 0 && HW <= 1024 && (reinterpret_cast<uintptr_t>(x) & 15) == 0)
    hipLaunchKernelGGL(plane_stats_small_kernel, dim3((B * C + 3) / 4), dim3(256), 0, s, x, coef, B * C, HW);
  else if (HW % 4 == 0 && HW <= 16384 && (reinterpret_cast<uintptr_t>(x) & 15) == 0)
    hipLaunchKernelGGL(plane_stats_kernel<true>, dim3(B * C), dim3(256), 0, s, x, coef, HW);
  else
    hipLaunchKernelGGL(plane_stats_kernel<false>, dim3(B * C), dim3(256), 0, s, x, coef, HW);
  hipLaunchKernelGGL(instnorm_plus_coef_kernel, dim3(B), dim3(256), 0, s, coef, alpha, gamma, beta, C, amax_bound,
                     sqrtf((float)HW));
  return ipdm_launch_status();
}

extern "C" int ipdm_instnorm_plus_coef_partials_f32(const float* partials, int P, const float* alpha, const float* gamma,
                                                    const float* beta, float* coef, int B, int C, int HW, float* amax_bound,
                                                    void* stream) {
  IPDM_REQUIRE(B >= 0 && C > 0 && P > 0 && (HW > 0 || !amax_bound));
  if (B == 0) return IPDM_OK;
  IPDM_REQUIRE(partials && alpha && gamma && coef);
  hipStream_t s = ipdm_stream(stream);
  hipLaunchKernelGGL(plane_stats_from_partials_kernel, dim3((B * C + 3) / 4), dim3(256), 0, s, partials, coef, B * C, P);
  hipLaunchKernelGGL(instnorm_plus_coef_kernel, dim3(B), dim3(256), 0, s, coef, alpha, gamma, beta, C, amax_bound,
                     sqrtf((float)(HW > 0 ? HW : 0)));
  return ipdm_launch_status();
}

extern "C" int ipdm_affine_act_f32(const float* x, const float* coef, float* y, int B, int C, int HW, int act,
                                   void* stream) {
  IPDM_REQUIRE(B >= 0 && C > 0 && HW > 0);
  if (B == 0) return IPDM_OK;
  IPDM_REQUIRE(x && coef && y);
  IPDM_REQUIRE((int64_t)B * C <= 65535 * 32);
  int gx = (HW / 4 + 255) / 256;
  if (gx < 1) gx = 1;
  if (gx > 64) gx = 64;
  int64_t planes = (int64_t)B * C;
  // gridDim.y is limited to 65535: fold planes in chunks
  for (int64_t p0 = 0; p0 < planes; p0 += 65535) {
    int np = (int)((planes - p0) < 65535 ? (planes - p0) : 65535);
    hipLaunchKernelGGL(affine_act_kernel, dim3(gx, np), dim3(256), 0, ipdm_stream(stream), x + p0 * HW, coef + p0 * 3,
                       y + p0 * HW, HW, act);
  }
  return ipdm_launch_status();
}

extern "C" int ipdm_affine_act_cat_f32(const float* x1, int C1, const float* x2, int C2, const float* coef, float* y, int B,
                                       int HW, int act, void* stream) {
  IPDM_REQUIRE(B >= 0 && C1 > 0 && C2 > 0 && HW > 0);
  if (B == 0) return IPDM_OK;
  IPDM_REQUIRE(x1 && x2 && coef && y && y != x1 && y != x2);
  const int64_t planes = (int64_t)B * (C1 + C2);
  if (planes > 65535) return IPDM_EUNSUPPORTED;                // one grid.y; the networks here stay far below it
  int gx = (HW / 4 + 255) / 256;
  gx = gx < 1 ? 1 : (gx > 64 ? 64 : gx);
  hipLaunchKernelGGL(affine_act_cat_kernel, dim3(gx, (unsigned)planes), dim3(256), 0, ipdm_stream(stream), x1, C1, x2, C2, coef,
                     y, HW, act);
  return ipdm_launch_status();
}

extern "C" int ipdm_act_f32(const float* x, float* y, int64_t n, int act, void* stream) {
  IPDM_REQUIRE(n >= 0);
  if (n == 0) return IPDM_OK;
  IPDM_REQUIRE(x && y);
  hipLaunchKernelGGL(ew1_kernel<ActOp>, dim3(ipdm_ew_grid(n / 4 + 1, 256)), dim3(256), 0, ipdm_stream(stream), x, y, n,
                     ActOp{act});
  return ipdm_launch_status();
}

extern "C" int ipdm_scale_shift_f32(const float* x, float* y, int64_t n, float a, float b, void* stream) {
  IPDM_REQUIRE(n >= 0);
  if (n == 0) return IPDM_OK;
  IPDM_REQUIRE(x && y);
  hipLaunchKernelGGL(ew1_kernel<ScaleShiftOp>, dim3(ipdm_ew_grid(n / 4 + 1, 256)), dim3(256), 0, ipdm_stream(stream), x,
                     y, n, ScaleShiftOp{a, b});
  return ipdm_launch_status();
}

extern "C" int ipdm_add_f32(const float* x, const float* y, float* out, int64_t n, void* stream) {
  IPDM_REQUIRE(n >= 0);
  if (n == 0) return IPDM_OK;
  IPDM_REQUIRE(x && y && out);
  hipLaunchKernelGGL(add_kernel, dim3(ipdm_ew_grid(n / 4 + 1, 256)), dim3(256), 0, ipdm_stream(stream), x, y, out, n);
  return ipdm_launch_status();
}

extern "C" int ipdm_div_sigma_f32(const float* x, const float* sigmas, const int64_t* labels, float* out, int B,
                                  int64_t sample_elems, void* stream) {
  IPDM_REQUIRE(B >= 0 && sample_elems >= 0 && B <= 65535);
  if (B == 0 || sample_elems == 0) return IPDM_OK;
  IPDM_REQUIRE(x && sigmas && out);
  int gx = (int)((sample_elems + 255) / 256);
  if (gx > 256) gx = 256;
  hipLaunchKernelGGL(div_sigma_kernel, dim3(gx, B), dim3(256), 0, ipdm_stream(stream), x, sigmas, labels, out,
                     (long long)sample_elems);
  return ipdm_launch_status();
}

extern "C" int ipdm_maxpool5_f32(const float* x, float* y, int planes, int H, int W, void* stream) {
  IPDM_REQUIRE(planes >= 0 && H > 0 && W > 0);
  if (planes == 0) return IPDM_OK;
  IPDM_REQUIRE(x && y && x != y);
  int tiles = ((H + MP_T - 1) / MP_T) * ((W + MP_T - 1) / MP_T);
  if (W % 4 == 0 && ((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(y)) & 15) == 0) {
    const int64_t n_items = (int64_t)planes * (W / 4) * ((H + MP_R - 1) / MP_R);
    if ((n_items + 255) / 256 <= 0x7fffffff) {
      hipLaunchKernelGGL(maxpool5_strip_kernel, dim3((unsigned)((n_items + 255) / 256)), dim3(256), 0, ipdm_stream(stream), x, y,
                         H, W, (long long)n_items);
      return ipdm_launch_status();
    }
  }
  for (int p0 = 0; p0 < planes; p0 += 65535) {
    int np = (planes - p0) < 65535 ? (planes - p0) : 65535;
    hipLaunchKernelGGL(maxpool5_kernel, dim3(tiles, np), dim3(256), 0, ipdm_stream(stream), x + (size_t)p0 * H * W,
                       y + (size_t)p0 * H * W, H, W);
  }
  return ipdm_launch_status();
}

extern "C" int ipdm_meanpool2_f32(const float* x, float* y, int planes, int H, int W, void* stream) {
  IPDM_REQUIRE(planes >= 0 && H > 0 && W > 0 && H % 2 == 0 && W % 2 == 0);
  if (planes == 0) return IPDM_OK;
  IPDM_REQUIRE(x && y);
  int64_t n_out = (int64_t)planes * (H / 2) * (W / 2);
  hipLaunchKernelGGL(meanpool2_kernel, dim3(ipdm_ew_grid(n_out, 256)), dim3(256), 0, ipdm_stream(stream), x, y,
                     (long long)n_out, H, W);
  return ipdm_launch_status();
}

// up-sampling form (in_w % 4 == 0, out_w % 4 == 0): one workgroup per (plane, strip of output rows) brings the few source
// rows the strip touches into LDS with float4 loads and gathers from there -- 16 scalar global loads per four outputs
// become a handful of wide ones (the gather was what held bilinear4_kernel at 2.8 TB/s).  Per-element arithmetic identical to
// bilinear_kernel.  LDS: (floor(sh*(RB-1)) + 3) * in_w floats, checked by the launcher.
constexpr int BL_RB = 16;                       // output rows per strip
__global__ __launch_bounds__(256) void bilinear_lds_kernel(const float* __restrict__ x, float* out, int ih, int iw, int oh,
                                                           int ow, float sh, float sw, int accumulate, int act, int lds_rows,
                                                           float* amax, int planes_per_image) {
  float amx = 0.f;
  extern __shared__ __align__(16) float bl_src[];
  const int64_t plane = blockIdx.y;
  const int r0 = blockIdx.x * BL_RB;
  const int r1 = min(oh, r0 + BL_RB);
  const int ys = (int)(sh * r0);                                     // first source row of the strip
  const int ye = min(ih - 1, (int)(sh * (r1 - 1)) + 1);              // last one
  const int nrows = ye - ys + 1;                                     // <= lds_rows (launcher)
  const float* sp = x + (plane * ih + ys) * (int64_t)iw;
  const int nq = nrows * iw / 4;
  for (int q = threadIdx.x; q < nq; q += 256)
    reinterpret_cast<float4*>(bl_src)[q] = reinterpret_cast<const float4*>(sp)[q];
  __syncthreads();
  const int og = ow / 4;
  for (int it = threadIdx.x; it < (r1 - r0) * og; it += 256) {
    const int ry = it / og, xg = it - ry * og;
    const int oy = r0 + ry;
    const float fy = sh * oy;
    const int y0 = (int)fy;
    const int yp = y0 < ih - 1 ? 1 : 0;
    const float ly1 = fy - y0, ly0 = 1.f - ly1;
    const float* q0 = bl_src + (y0 - ys) * iw;
    const float* q1 = q0 + yp * iw;
    float* po = out + (plane * oh + oy) * (int64_t)ow + xg * 4;
    const float4 acc = accumulate ? *reinterpret_cast<const float4*>(po) : make_float4(0.f, 0.f, 0.f, 0.f);
    float res[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const float fx = sw * (xg * 4 + j);
      const int x0 = (int)fx;
      const int xp = x0 < iw - 1 ? 1 : 0;
      const float lx1 = fx - x0, lx0 = 1.f - lx1;
      res[j] = ly0 * (lx0 * q0[x0] + lx1 * q0[x0 + xp]) + ly1 * (lx0 * q1[x0] + lx1 * q1[x0 + xp]);
    }
    float4 o;
    o.x = ipdm_act(accumulate ? acc.x + res[0] : res[0], act);
    o.y = ipdm_act(accumulate ? acc.y + res[1] : res[1], act);
    o.z = ipdm_act(accumulate ? acc.z + res[2] : res[2], act);
    o.w = ipdm_act(accumulate ? acc.w + res[3] : res[3], act);
    amx = fmaxf(fmaxf(amx, fmaxf(fabsf(o.x), fabsf(o.y))), fmaxf(fabsf(o.z), fabsf(o.w)));
    *reinterpret_cast<float4*>(po) = o;
  }
  if (amax) amax_commit(amx, amax + (size_t)(plane / planes_per_image) * IPDM_AMAX_SLOT, (int)(blockIdx.x + blockIdx.y));
}

extern "C" int ipdm_bilinear_f32(const float* x, float* out, int planes, int in_h, int in_w, int out_h, int out_w,
                                 int accumulate, int act, int planes_per_image, float* amax_out, void* stream) {
  IPDM_REQUIRE(planes >= 0 && in_h > 0 && in_w > 0 && out_h > 0 && out_w > 0);
  if (planes == 0) return IPDM_OK;
  IPDM_REQUIRE(x && out);
  IPDM_REQUIRE(!amax_out || (planes_per_image > 0 && planes % planes_per_image == 0 && planes / planes_per_image <= 65535));
  const int n_img = amax_out ? planes / planes_per_image : 1;       // grid.y of the grid-stride forms
  const int ppi = amax_out ? planes_per_image : 1;
  float sh = out_h > 1 ? (float)(in_h - 1) / (float)(out_h - 1) : 0.f;
  float sw = out_w > 1 ? (float)(in_w - 1) / (float)(out_w - 1) : 0.f;
  int64_t n_out = (int64_t)planes * out_h * out_w;
  const bool aligned = ((reinterpret_cast<uintptr_t>(out) | reinterpret_cast<uintptr_t>(x)) & 15) == 0;
  const int lds_rows = (int)(sh * (BL_RB - 1)) + 3;
  if (aligned && out_w % 4 == 0 && in_w % 4 == 0 && out_h >= in_h && planes <= 65535 &&
      (size_t)lds_rows * in_w * sizeof(float) <= 48 * 1024) {
    hipLaunchKernelGGL(bilinear_lds_kernel, dim3((out_h + BL_RB - 1) / BL_RB, planes), dim3(256),
                       (size_t)lds_rows * in_w * sizeof(float), ipdm_stream(stream), x, out, in_h, in_w, out_h, out_w, sh, sw,
                       accumulate, act, lds_rows, amax_out, ppi);
    return ipdm_launch_status();
  }
  auto grid = [&](int64_t per_slice) {                               // ~2048 workgroups in all
    int64_t gx = (per_slice + 255) / 256, cap = (2048 + n_img - 1) / n_img;
    gx = gx < 1 ? 1 : (gx > cap ? cap : gx);
    return dim3((unsigned)gx, (unsigned)n_img);
  };
  if (out_w % 4 == 0 && (reinterpret_cast<uintptr_t>(out) & 15) == 0) {
    const int64_t per = n_out / 4 / n_img;
    hipLaunchKernelGGL(bilinear4_kernel, grid(per), dim3(256), 0, ipdm_stream(stream), x, out, (long long)per, in_h, in_w,
                       out_h, out_w, sh, sw, accumulate, act, amax_out);
    return ipdm_launch_status();
  }
  hipLaunchKernelGGL(bilinear_kernel, grid(n_out / n_img), dim3(256), 0, ipdm_stream(stream), x, out, (long long)(n_out / n_img),
                     in_h, in_w, out_h, out_w, sh, sw, accumulate, act, amax_out);
  return ipdm_launch_status();
}

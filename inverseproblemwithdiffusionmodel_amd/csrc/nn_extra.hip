// Small kernels the NCSN++ score network and the predictor-corrector samplers need besides the shared
// convolution / resampling kernels: GroupNorm coefficients, Linear, spatial self-attention, a*x + b*y,
// per-sample updates and norms.  All launch/HBM-bound; none of them is on the ACDC headline path.
//
// Reference semantics: torch.nn.GroupNorm(eps 1e-6) as used in models/layerspp.py:66,219 and models/ncsnpp.py:194;
// torch.nn.Linear (models/ncsnpp.py:85-90, layerspp.py:223); AttnBlockpp.forward (models/layerspp.py:75-91);
// the update arithmetic of ReverseDiffusionPredictor / LangevinCorrector (sde/sampling.py:200-205, 267-287).
#include "ipdm_common.h"

namespace {

// one workgroup per (image, group): mean / biased variance over the group's (C/G)*HW elements (two sweeps),
// then coef[b][c] = (mu, weight_c * rstd, bias_c) for the channels of the group
__global__ __launch_bounds__(256) void groupnorm_coef_kernel(const float* __restrict__ x, const float* __restrict__ weight,
                                                             const float* __restrict__ bias, float* __restrict__ coef,
                                                             int C, int HW, int G, float eps) {
  __shared__ double red[4];
  const int b = blockIdx.x / G, g = blockIdx.x % G;
  const int cpg = C / G;
  const int64_t n = (int64_t)cpg * HW;
  const float* p = x + ((size_t)b * C + (size_t)g * cpg) * HW;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  float s = 0.f;
  for (int64_t i = tid; i < n; i += 256) s += p[i];
  double ds = ipdm_wave_sum((double)s);
  if (lane == 0) red[wave] = ds;
  __syncthreads();
  const float mean = (float)((red[0] + red[1] + red[2] + red[3]) / (double)n);
  __syncthreads();
  float q = 0.f;
  for (int64_t i = tid; i < n; i += 256) {
    float d = p[i] - mean;
    q += d * d;
  }
  double dq = ipdm_wave_sum((double)q);
  if (lane == 0) red[wave] = dq;
  __syncthreads();
  const float var = (float)((red[0] + red[1] + red[2] + red[3]) / (double)n);
  const float rstd = 1.0f / sqrtf(var + eps);
  for (int c = tid; c < cpg; c += 256) {
    const int ch = g * cpg + c;
    float* o = coef + ((size_t)b * C + ch) * 3;
    o[0] = mean;
    o[1] = (weight ? weight[ch] : 1.f) * rstd;
    o[2] = bias ? bias[ch] : 0.f;
  }
}

// y[b][o] = bias[o] + sum_i act(x[b][i]) * W[o][i]     (torch.nn.Linear layout), one wave per output
// GroupNorm in two register-resident passes (planes of up to 65536 elements): x is read ONCE.
//   gn_plane_kernel<T>: one workgroup of T threads per (image, channel) plane, 16 float4 per thread:
//       mean_c and M2_c = sum (x - mean_c)^2, parked in coef[b][c][0..1]
//   gn_combine_kernel:  one workgroup per (image, group): equal-sized planes combine exactly as
//       mean_g = avg(mean_c),  M2_g = sum M2_c + HW * sum (mean_c - mean_g)^2   (Chan et al.)
// against (B x G) workgroups sweeping (C/G)*HW elements twice in groupnorm_coef_kernel (kept for larger planes).
// (c_src, c_tot, c_off): the planes of `x` are channels c_off .. c_off + c_src - 1 of a (virtual) tensor of c_tot channels whose
// coefficients `coef` holds -- GroupNorm over torch.cat([x1, x2], dim=1) without the concatenation; (C, C, 0) otherwise
// plane_amax (may be NULL): max |x| of the plane, same slot -- the f16x2 convolutions' dynamic range for free where the tensor is
// normalised anyway (the plane is in registers)
template <int T>
__global__ __launch_bounds__(T) void gn_plane_kernel(const float* __restrict__ x, float* __restrict__ coef, int HW, int c_src,
                                                     int c_tot, int c_off, float* __restrict__ plane_amax) {
  __shared__ double red[T / 64];
  const float* p = x + (size_t)blockIdx.x * HW;
  const size_t slot = (size_t)(blockIdx.x / c_src) * c_tot + c_off + blockIdx.x % c_src;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int n4 = HW / 4;
  float4 keep[16];
  float s = 0.f;
#pragma unroll
  for (int k = 0; k < 16; ++k) {
    const int i = tid + k * T;
    keep[k] = i < n4 ? reinterpret_cast<const float4*>(p)[i] : make_float4(0.f, 0.f, 0.f, 0.f);
  }
#pragma unroll
  for (int k = 0; k < 16; ++k)
    if (tid + k * T < n4) s += (keep[k].x + keep[k].y) + (keep[k].z + keep[k].w);
  double ds = ipdm_wave_sum((double)s);
  if (lane == 0) red[wave] = ds;
  __syncthreads();
  double tot = 0.0;
#pragma unroll
  for (int w = 0; w < T / 64; ++w) tot += red[w];
  const float mean = (float)(tot / (double)HW);
  __syncthreads();
  float q = 0.f, am = 0.f;
#pragma unroll
  for (int k = 0; k < 16; ++k) {
    if (tid + k * T < n4) {
      const float a = keep[k].x - mean, b = keep[k].y - mean, c = keep[k].z - mean, d = keep[k].w - mean;
      q += (a * a + b * b) + (c * c + d * d);
      am = fmaxf(fmaxf(am, fmaxf(fabsf(keep[k].x), fabsf(keep[k].y))), fmaxf(fabsf(keep[k].z), fabsf(keep[k].w)));
    }
  }
  if (plane_amax) {                                            // (uniform branch)
    __shared__ float redm[T / 64];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) am = fmaxf(am, __shfl_xor(am, o, 64));
    if (lane == 0) redm[wave] = am;
    __syncthreads();
    if (tid == 0) {
      float m = 0.f;
#pragma unroll
      for (int w = 0; w < T / 64; ++w) m = fmaxf(m, redm[w]);
      plane_amax[slot] = m;
    }
  }
  double dq = ipdm_wave_sum((double)q);
  if (lane == 0) red[wave] = dq;
  __syncthreads();
  if (tid == 0) {
    double m2 = 0.0;
#pragma unroll
    for (int w = 0; w < T / 64; ++w) m2 += red[w];
    coef[slot * 3 + 0] = mean;
    coef[slot * 3 + 1] = (float)m2;
  }
}

// Whole GROUPS of up to 65536 elements (every level of NCSN++ below the full resolution): the cpg planes of a group are
// contiguous in NCHW, so the group is ONE register-resident block -- mean and variance in two in-register sweeps exactly as the
// definition reads, and the workgroup writes the coefficients of its cpg channels itself: one launch instead of plane + combine
// (round 3: 109 gn_combine launches of 4.8 us per NCSN++ forward).  Two sources (x1 holds channels [0, C1), x2 the rest of the
// virtual concatenation; C1 a multiple of the group's channel count, so a group never straddles them; x2 == NULL: one source).
template <int T>
__global__ __launch_bounds__(T) void gn_group_kernel(const float* __restrict__ x1, int C1, const float* __restrict__ x2, int C2,
                                                     const float* __restrict__ weight, const float* __restrict__ bias,
                                                     float* __restrict__ coef, int HW, int G, float eps) {
  __shared__ double red[T / 64];
  const int C = C1 + C2, cpg = C / G;
  const int b = blockIdx.x / G, g = blockIdx.x % G;
  const int c0 = g * cpg;
  const float* p = c0 < C1 ? x1 + ((size_t)b * C1 + c0) * HW : x2 + ((size_t)b * C2 + (c0 - C1)) * HW;
  const int n = cpg * HW, n4 = n / 4;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  float4 keep[16];
  float s = 0.f;
#pragma unroll
  for (int k = 0; k < 16; ++k) {
    const int i = tid + k * T;
    keep[k] = i < n4 ? reinterpret_cast<const float4*>(p)[i] : make_float4(0.f, 0.f, 0.f, 0.f);
  }
#pragma unroll
  for (int k = 0; k < 16; ++k)
    if (tid + k * T < n4) s += (keep[k].x + keep[k].y) + (keep[k].z + keep[k].w);
  double ds = ipdm_wave_sum((double)s);
  if (lane == 0) red[wave] = ds;
  __syncthreads();
  double tot = 0.0;
#pragma unroll
  for (int w = 0; w < T / 64; ++w) tot += red[w];
  const float mean = (float)(tot / (double)n);
  __syncthreads();
  float q = 0.f;
#pragma unroll
  for (int k = 0; k < 16; ++k) {
    if (tid + k * T < n4) {
      const float a = keep[k].x - mean, bb = keep[k].y - mean, c = keep[k].z - mean, d = keep[k].w - mean;
      q += (a * a + bb * bb) + (c * c + d * d);
    }
  }
  double dq = ipdm_wave_sum((double)q);
  if (lane == 0) red[wave] = dq;
  __syncthreads();
  double m2 = 0.0;
#pragma unroll
  for (int w = 0; w < T / 64; ++w) m2 += red[w];
  const float rstd = 1.0f / sqrtf((float)(m2 / (double)n) + eps);
  for (int c = tid; c < cpg; c += T) {
    const int ch = c0 + c;
    float* o = coef + ((size_t)b * C + ch) * 3;
    o[0] = mean;
    o[1] = (weight ? weight[ch] : 1.f) * rstd;
    o[2] = bias ? bias[ch] : 0.f;
  }
}

__global__ __launch_bounds__(64) void gn_combine_kernel(const float* __restrict__ weight, const float* __restrict__ bias,
                                                        float* __restrict__ coef, int C, int HW, int G, float eps) {
  const int b = blockIdx.x / G, g = blockIdx.x % G;
  const int cpg = C / G;
  float* cg = coef + ((size_t)b * C + (size_t)g * cpg) * 3;
  const int lane = threadIdx.x;
  double sm = 0.0, s2 = 0.0;
  for (int c = lane; c < cpg; c += 64) {
    sm += (double)cg[c * 3 + 0];
    s2 += (double)cg[c * 3 + 1];
  }
  sm = ipdm_wave_sum(sm);
  s2 = ipdm_wave_sum(s2);
  const double mean_g = sm / (double)cpg;
  double dev = 0.0;
  for (int c = lane; c < cpg; c += 64) {
    const double dm = (double)cg[c * 3 + 0] - mean_g;
    dev += dm * dm;
  }
  dev = ipdm_wave_sum(dev);
  const double var = (s2 + (double)HW * dev) / ((double)cpg * (double)HW);
  const float rstd = 1.0f / sqrtf((float)var + eps);
  const float mg = (float)mean_g;
  // all reads of the parked (mean_c, M2_c) are done (wave-synchronous: one wave per workgroup)
  for (int c = lane; c < cpg; c += 64) {
    const int ch = g * cpg + c;
    cg[c * 3 + 0] = mg;
    cg[c * 3 + 1] = (weight ? weight[ch] : 1.f) * rstd;
    cg[c * 3 + 2] = bias ? bias[ch] : 0.f;
  }
}

__global__ __launch_bounds__(256) void linear_kernel(const float* __restrict__ x, const float* __restrict__ W,
                                                     const float* __restrict__ bias, float* __restrict__ y, int B,
                                                     int In, int Out, int act) {
  const int wave = (blockIdx.x * 256 + threadIdx.x) >> 6, lane = threadIdx.x & 63;
  if (wave >= B * Out) return;
  const int b = wave / Out, o = wave % Out;
  float s = 0.f;
  for (int i = lane; i < In; i += 64) s += ipdm_act(x[(size_t)b * In + i], act) * W[(size_t)o * In + i];
  s = ipdm_wave_sum(s);
  if (lane == 0) y[(size_t)b * Out + o] = s + (bias ? bias[o] : 0.f);
}

// spatial self-attention over N = H*W positions with C channels, q/k/v/out [B][C][N]:
//   out[:, i] = sum_j softmax_j(q[:, i] . k[:, j] * scale) v[:, j]
// one workgroup per (image, query); N <= 1024, C <= 1024 (attention sits at 16x16 in every shipped config)
__global__ __launch_bounds__(256) void attention_kernel(const float* __restrict__ q, const float* __restrict__ k,
                                                        const float* __restrict__ v, float* __restrict__ out, int C,
                                                        int N, float scale) {
  extern __shared__ float sm[];
  float* qs = sm;            // [C]
  float* ps = sm + C;        // [N]
  __shared__ float red[4];
  const int b = blockIdx.y, i = blockIdx.x;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const float* qb = q + (size_t)b * C * N;
  const float* kb = k + (size_t)b * C * N;
  const float* vb = v + (size_t)b * C * N;
  for (int c = tid; c < C; c += 256) qs[c] = qb[(size_t)c * N + i];
  __syncthreads();
  float mx = -INFINITY;
  for (int j = tid; j < N; j += 256) {
    float s = 0.f;
    for (int c = 0; c < C; ++c) s += qs[c] * kb[(size_t)c * N + j];
    s *= scale;
    ps[j] = s;
    mx = fmaxf(mx, s);
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o, 64));
  if (lane == 0) red[wave] = mx;
  __syncthreads();
  mx = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
  __syncthreads();
  float sum = 0.f;
  for (int j = tid; j < N; j += 256) {
    float e = expf(ps[j] - mx);
    ps[j] = e;
    sum += e;
  }
  sum = ipdm_wave_sum(sum);
  if (lane == 0) red[wave] = sum;
  __syncthreads();
  const float inv = 1.0f / (red[0] + red[1] + red[2] + red[3]);
  for (int c = wave; c < C; c += 4) {            // one wave per channel row of v: coalesced over j
    float s = 0.f;
    for (int j = lane; j < N; j += 64) s += ps[j] * vb[(size_t)c * N + j];
    s = ipdm_wave_sum(s);
    if (lane == 0) out[((size_t)b * C + c) * N + i] = s * inv;
  }
}

__global__ __launch_bounds__(256) void axpby_kernel(const float* x, const float* y, float* out, int64_t n, float a,
                                                    float b) {
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256)
    out[i] = a * x[i] + b * y[i];
}

// out[s][:] = x[s][:] + a[s]*y[s][:] + c[s]*z[s][:]   (per-sample coefficients in device memory; z may be NULL)
__global__ __launch_bounds__(256) void sample_axpy2_kernel(const float* x, const float* __restrict__ y,
                                                           const float* __restrict__ z, const float* __restrict__ a,
                                                           const float* __restrict__ c, float* out, int64_t elems) {
  const int s = blockIdx.y;
  const float as = a[s], cs = (z && c) ? c[s] : 0.f;
  const size_t base = (size_t)s * elems;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < elems; i += (int64_t)gridDim.x * 256) {
    float v = x[base + i] + as * y[base + i];
    if (z) v += cs * z[base + i];
    out[base + i] = v;
  }
}

// norms[s] = ||x[s]||_2, one workgroup per sample
__global__ __launch_bounds__(256) void sample_norm_kernel(const float* __restrict__ x, float* __restrict__ norms,
                                                          int64_t elems) {
  __shared__ double red[4];
  const float* p = x + (size_t)blockIdx.x * elems;
  float q = 0.f;
  for (int64_t i = threadIdx.x; i < elems; i += 256) q += p[i] * p[i];
  double dq = ipdm_wave_sum((double)q);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = dq;
  __syncthreads();
  if (threadIdx.x == 0) norms[blockIdx.x] = (float)sqrt(red[0] + red[1] + red[2] + red[3]);
}

// torch.optim.Adam (no weight decay, no amsgrad), one step in place on x with the ASCENT direction g (the reference
// hands Adam param.grad = -grad, MAP_optimizers.py:103-104):  m = b1 m + (1-b1)(-g);  v = b2 v + (1-b2) g^2;
// x -= (lr / (1 - b1^t)) * m / (sqrt(v) / sqrt(1 - b2^t) + eps)
__global__ __launch_bounds__(256) void adam_ascent_kernel(float* __restrict__ x, const float* __restrict__ g,
                                                          float* __restrict__ m, float* __restrict__ v, int64_t n,
                                                          float b1, float b2, float step_size, float sqrt_bc2,
                                                          float eps) {
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
    const float gr = -g[i];
    const float mi = m[i] + (gr - m[i]) * (1.f - b1);        // lerp, as torch
    const float vi = v[i] * b2 + (1.f - b2) * (gr * gr);
    m[i] = mi;
    v[i] = vi;
    x[i] = x[i] - step_size * (mi / (sqrtf(vi) / sqrt_bc2 + eps));
  }
}

}  // namespace

extern "C" int ipdm_adam_ascent_f32(float* x, const float* g, float* m, float* v, int64_t n, float lr, float beta1,
                                    float beta2, float eps, int step, void* stream) {
  IPDM_REQUIRE(n >= 0 && step >= 1 && lr > 0.f);
  if (n == 0) return IPDM_OK;
  IPDM_REQUIRE(x && g && m && v);
  const double bc1 = 1.0 - pow((double)beta1, (double)step), bc2 = 1.0 - pow((double)beta2, (double)step);
  hipLaunchKernelGGL(adam_ascent_kernel, dim3(ipdm_ew_grid(n, 256)), dim3(256), 0, ipdm_stream(stream), x, g, m, v,
                     (long long)n, beta1, beta2, (float)((double)lr / bc1), (float)sqrt(bc2), eps);
  return ipdm_launch_status();
}

// GroupNorm coefficients from the producing convolutions' statistics partials ([B][C][P][3] = count, mean, sum of squared deviations
// per pixel block; conv_wino1d.hip / conv_wino_bx3.hip STATS epilogues) -- the tensor is not read at all.  One 256-thread workgroup
// per (image, group): thread t merges the group's partials t, t+256, ... in index order (Chan et al., float64), a butterfly merges
// the lanes of a wave and every thread then merges the four waves' triples in wave order (symmetric updates: a fixed tree, whatever
// the batch).  (part2, C2): the group's channels >= C1 come from a second tensor's
// partials -- GroupNorm of torch.cat([x1, x2], dim=1) without the concatenation.
__global__ __launch_bounds__(256) void gn_from_partials_kernel(const float* __restrict__ part1, int C1, const float* __restrict__ part2,
                                                              int C2, int P, const float* __restrict__ weight,
                                                              const float* __restrict__ bias, float* __restrict__ coef, int G,
                                                              float eps) {
  const int C = C1 + C2, cpg = C / G;
  const int b = blockIdx.x / G, g = blockIdx.x % G;
  const int lane = threadIdx.x;
  __shared__ double red[4][3];
  double n = 0.0, mean = 0.0, m2 = 0.0;
  for (int i = lane; i < cpg * P; i += 256) {
    const int c = g * cpg + i / P, q = i % P;
    const float* pp = c < C1 ? part1 + (((size_t)b * C1 + c) * P + q) * 3 : part2 + (((size_t)b * C2 + (c - C1)) * P + q) * 3;
    const double nb = pp[0], mb = pp[1], qb = pp[2];
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
      const double d = mb - mean;
      const double new_mean = (n * mean + nb * mb) / nn;
      m2 = (m2 + qb) + d * d * (n * nb / nn);
      mean = new_mean;
      n = nn;
    }
  }
  if ((lane & 63) == 0) {
    red[lane >> 6][0] = n; red[lane >> 6][1] = mean; red[lane >> 6][2] = m2;
  }
  __syncthreads();
  n = red[0][0]; mean = red[0][1]; m2 = red[0][2];
#pragma unroll
  for (int w = 1; w < 4; ++w) {
    const double nb = red[w][0], mb = red[w][1], qb = red[w][2];
    const double nn = n + nb;
    if (nn > 0.0) {
      const double d = mb - mean;
      const double new_mean = (n * mean + nb * mb) / nn;
      m2 = (m2 + qb) + d * d * (n * nb / nn);
      mean = new_mean;
      n = nn;
    }
  }
  const float rstd = 1.0f / sqrtf((float)(n > 0.0 ? m2 / n : 0.0) + eps);
  for (int c = lane; c < cpg; c += 256) {
    const int ch = g * cpg + c;
    float* o = coef + ((size_t)b * C + ch) * 3;
    o[0] = (float)mean;
    o[1] = (weight ? weight[ch] : 1.f) * rstd;
    o[2] = bias ? bias[ch] : 0.f;
  }
}

extern "C" int ipdm_groupnorm_coef_partials_f32(const float* part1, int C1, const float* part2, int C2, int P, const float* weight,
                                                const float* bias, float* coef, int B, int G, float eps, void* stream) {
  IPDM_REQUIRE(B >= 0 && C1 > 0 && C2 >= 0 && P > 0 && G > 0 && (C1 + C2) % G == 0);
  if (B == 0) return IPDM_OK;
  IPDM_REQUIRE(part1 && coef && (C2 == 0 || part2));
  hipLaunchKernelGGL(gn_from_partials_kernel, dim3(B * G), dim3(256), 0, ipdm_stream(stream), part1, C1, part2, C2, P, weight, bias,
                     coef, G, eps);
  return ipdm_launch_status();
}

extern "C" int ipdm_groupnorm_coef_f32(const float* x, const float* weight, const float* bias, float* coef, int B, int C,
                                       int HW, int G, float eps, float* plane_amax, void* stream) {
  IPDM_REQUIRE(B >= 0 && C > 0 && HW > 0 && G > 0 && C % G == 0);
  if (B == 0) return IPDM_OK;
  IPDM_REQUIRE(x && coef);
  const int64_t gn = (int64_t)(C / G) * HW;                   // elements of a group
  if (!plane_amax && gn % 4 == 0 && gn <= 65536 && (reinterpret_cast<uintptr_t>(x) & 15) == 0) {
    if (gn <= 16384)
      hipLaunchKernelGGL(gn_group_kernel<256>, dim3(B * G), dim3(256), 0, ipdm_stream(stream), x, C, nullptr, 0, weight, bias, coef,
                         HW, G, eps);
    else
      hipLaunchKernelGGL(gn_group_kernel<1024>, dim3(B * G), dim3(1024), 0, ipdm_stream(stream), x, C, nullptr, 0, weight, bias,
                         coef, HW, G, eps);
    return ipdm_launch_status();
  }
  if (HW % 4 == 0 && HW <= 65536 && (reinterpret_cast<uintptr_t>(x) & 15) == 0) {
    if (HW <= 16384)
      hipLaunchKernelGGL(gn_plane_kernel<256>, dim3(B * C), dim3(256), 0, ipdm_stream(stream), x, coef, HW, C, C, 0, plane_amax);
    else
      hipLaunchKernelGGL(gn_plane_kernel<1024>, dim3(B * C), dim3(1024), 0, ipdm_stream(stream), x, coef, HW, C, C, 0, plane_amax);
    hipLaunchKernelGGL(gn_combine_kernel, dim3(B * G), dim3(64), 0, ipdm_stream(stream), weight, bias, coef, C, HW, G, eps);
    return ipdm_launch_status();
  }
  if (plane_amax) return IPDM_EUNSUPPORTED;                   // the two-sweep fallback does not carry the maxima
  hipLaunchKernelGGL(groupnorm_coef_kernel, dim3(B * G), dim3(256), 0, ipdm_stream(stream), x, weight, bias, coef, C, HW,
                     G, eps);
  return ipdm_launch_status();
}

// GroupNorm coefficients of torch.cat([x1, x2], dim=1) without the concatenation (single-read plane kernels only: planes of
// whole float4s up to 256 x 256; IPDM_EUNSUPPORTED otherwise -- the caller concatenates and uses ipdm_groupnorm_coef_f32)
extern "C" int ipdm_groupnorm_coef_cat_f32(const float* x1, int C1, const float* x2, int C2, const float* weight,
                                           const float* bias, float* coef, int B, int HW, int G, float eps, float* plane_amax,
                                           void* stream) {
  const int C = C1 + C2;
  IPDM_REQUIRE(B >= 0 && C1 > 0 && C2 > 0 && HW > 0 && G > 0 && C % G == 0);
  if (B == 0) return IPDM_OK;
  IPDM_REQUIRE(x1 && x2 && coef);
  if (!(HW % 4 == 0 && HW <= 65536 && ((reinterpret_cast<uintptr_t>(x1) | reinterpret_cast<uintptr_t>(x2)) & 15) == 0))
    return IPDM_EUNSUPPORTED;
  hipStream_t s = ipdm_stream(stream);
  const int cpg = C / G;
  const int64_t gn = (int64_t)cpg * HW;
  if (!plane_amax && C1 % cpg == 0 && gn <= 65536) {           // whole groups in registers: one launch
    if (gn <= 16384)
      hipLaunchKernelGGL(gn_group_kernel<256>, dim3(B * G), dim3(256), 0, s, x1, C1, x2, C2, weight, bias, coef, HW, G, eps);
    else
      hipLaunchKernelGGL(gn_group_kernel<1024>, dim3(B * G), dim3(1024), 0, s, x1, C1, x2, C2, weight, bias, coef, HW, G, eps);
    return ipdm_launch_status();
  }
  if (HW <= 16384) {
    hipLaunchKernelGGL(gn_plane_kernel<256>, dim3(B * C1), dim3(256), 0, s, x1, coef, HW, C1, C, 0, plane_amax);
    hipLaunchKernelGGL(gn_plane_kernel<256>, dim3(B * C2), dim3(256), 0, s, x2, coef, HW, C2, C, C1, plane_amax);
  } else {
    hipLaunchKernelGGL(gn_plane_kernel<1024>, dim3(B * C1), dim3(1024), 0, s, x1, coef, HW, C1, C, 0, plane_amax);
    hipLaunchKernelGGL(gn_plane_kernel<1024>, dim3(B * C2), dim3(1024), 0, s, x2, coef, HW, C2, C, C1, plane_amax);
  }
  hipLaunchKernelGGL(gn_combine_kernel, dim3(B * G), dim3(64), 0, s, weight, bias, coef, C, HW, G, eps);
  return ipdm_launch_status();
}

extern "C" int ipdm_linear_f32(const float* x, const float* W, const float* bias, float* y, int B, int In, int Out,
                               int act, void* stream) {
  IPDM_REQUIRE(B >= 0 && In > 0 && Out > 0);
  if (B == 0) return IPDM_OK;
  IPDM_REQUIRE(x && W && y);
  const int64_t waves = (int64_t)B * Out;
  hipLaunchKernelGGL(linear_kernel, dim3((unsigned)((waves + 3) / 4)), dim3(256), 0, ipdm_stream(stream), x, W, bias, y,
                     B, In, Out, act);
  return ipdm_launch_status();
}

extern "C" int ipdm_attention_f32(const float* q, const float* k, const float* v, float* out, int B, int C, int N,
                                  float scale, void* stream) {
  IPDM_REQUIRE(B >= 0 && C > 0 && N > 0);
  if (B == 0) return IPDM_OK;
  IPDM_REQUIRE(q && k && v && out);
  if (N > 4096 || C > 4096 || B > 65535) return IPDM_EUNSUPPORTED;
  hipLaunchKernelGGL(attention_kernel, dim3(N, B), dim3(256), (size_t)(C + N) * sizeof(float), ipdm_stream(stream), q, k,
                     v, out, C, N, scale);
  return ipdm_launch_status();
}

extern "C" int ipdm_axpby_f32(const float* x, const float* y, float* out, int64_t n, float a, float b, void* stream) {
  IPDM_REQUIRE(n >= 0);
  if (n == 0) return IPDM_OK;
  IPDM_REQUIRE(x && y && out);
  hipLaunchKernelGGL(axpby_kernel, dim3(ipdm_ew_grid(n, 256)), dim3(256), 0, ipdm_stream(stream), x, y, out, (long long)n,
                     a, b);
  return ipdm_launch_status();
}

extern "C" int ipdm_sample_axpy2_f32(const float* x, const float* y, const float* z, const float* a, const float* c,
                                     float* out, int n_samples, int64_t sample_elems, void* stream) {
  IPDM_REQUIRE(n_samples >= 0 && sample_elems >= 0 && n_samples <= 65535);
  if (n_samples == 0 || sample_elems == 0) return IPDM_OK;
  IPDM_REQUIRE(x && y && a && out);
  int gx = (int)((sample_elems + 255) / 256);
  if (gx > 512) gx = 512;
  hipLaunchKernelGGL(sample_axpy2_kernel, dim3(gx, n_samples), dim3(256), 0, ipdm_stream(stream), x, y, z, a, c, out,
                     (long long)sample_elems);
  return ipdm_launch_status();
}

extern "C" int ipdm_sample_norm_f32(const float* x, float* norms, int n_samples, int64_t sample_elems, void* stream) {
  IPDM_REQUIRE(n_samples >= 0 && sample_elems >= 0);
  if (n_samples == 0) return IPDM_OK;
  IPDM_REQUIRE(x && norms);
  hipLaunchKernelGGL(sample_norm_kernel, dim3(n_samples), dim3(256), 0, ipdm_stream(stream), x, norms,
                     (long long)sample_elems);
  return ipdm_launch_status();
}

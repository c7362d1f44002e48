// k-space operators for images that do not fit one CU's LDS (power-of-two sizes above 128x128, e.g. the 256x256 ACDC
// slices of the reference's real-data front end, helpers/load_data.py:274): the centred 2-D FFT runs as a ROW pass and a
// COLUMN pass over 64 KiB strips held in LDS, with the operator's elementwise work fused into the loads / stores of the
// passes and -- for the proximal operators -- the forward and the inverse column transform of a strip back to back in
// one kernel (the masked residual never leaves LDS; column strips without a sampled line skip both transforms, which at
// R = 40 is most of them).  The multi-coil proximal / Langevin step becomes four launches
//     Langevin update (planar, in place)  ->  rows forward, all coils  ->  columns forward + residual + columns inverse
//     ->  rows inverse + S_c-weighted coil sum (registers, coil order: deterministic) + update
// through a workspace of n_coils x B images (ipdm_sense_workspace_bytes).  Same arithmetic conventions as kspace.hip:
// fftshift / ifftshift folded into (-1)^(r+c) sign flips, orthonormal scale 1/sqrt(HW) applied once per 2-D transform.
#include "kspace_fft.h"

namespace ipdm_kspace_large {

using namespace ipdm_kspace;

constexpr int STRIP_ELEMS = 8192;                    // complex elements per workgroup strip (64 KiB)
constexpr int EPT = STRIP_ELEMS / FFT_THREADS;       // 8 elements per thread

bool large_ok(int H, int W) {
  return is_pow2(H) && is_pow2(W) && H >= 4 && W >= 4 && H <= 2048 && W <= 2048 && (int64_t)H * W > FFT_MAX_ELEMS;
}
static inline size_t strip_lds_bytes(int n) { return ((size_t)STRIP_ELEMS + (size_t)n) * sizeof(float2); }

#define STRIP_LDS_SETUP(N)                                    \
  extern __shared__ __align__(16) unsigned char smem_raw[];  \
  FftLds L;                                                   \
  L.buf = reinterpret_cast<float2*>(smem_raw);               \
  L.tw = L.buf + STRIP_ELEMS;                                \
  L.twN = (N);                                               \
  fft_make_twiddles(L);

// ---- generic passes ---------------------------------------------------------------------------------------------
// F: load(b, coil, r, c) -> float2 and store(b, coil, r, c, v); rows [r0, r0 + RS) of image (b, coil)
template <class F>
__global__ __launch_bounds__(FFT_THREADS) void rows_kernel(F f, int H, int W, int inverse) {
  STRIP_LDS_SETUP(W)
  const int RS = min(H, STRIP_ELEMS / W);
  const int r0 = blockIdx.x * RS, b = blockIdx.y, coil = blockIdx.z;
  const int n = RS * W;
  for (int e = threadIdx.x; e < n; e += FFT_THREADS) {
    const int lr = e / W, c = e - lr * W;
    L.buf[e] = f.load(b, coil, r0 + lr, c);
  }
  __syncthreads();
  fft_lines(L, W, 1, W, RS, false, inverse != 0);
  for (int e = threadIdx.x; e < n; e += FFT_THREADS) {
    const int lr = e / W, c = e - lr * W;
    f.store(b, coil, r0 + lr, c, L.buf[e]);
  }
}

// columns [c0, c0 + CS) of image (b, coil); LDS layout buf[r * CS + lc].  With TWO_WAY the strip is transformed forward,
// F::mid is applied in place, and it is transformed back (inverse) before the store; F::skip lets a strip whose mid()
// is identically zero bypass both transforms.
template <class F, bool TWO_WAY>
__global__ __launch_bounds__(FFT_THREADS) void cols_kernel(F f, int H, int W, int inverse) {
  STRIP_LDS_SETUP(H)
  const int CS = min(W, STRIP_ELEMS / H);
  const int c0 = blockIdx.x * CS, b = blockIdx.y, coil = blockIdx.z;
  const int n = H * CS;
  if constexpr (TWO_WAY) {
    if (f.skip(b, c0, CS)) {                                   // uniform over the workgroup
      for (int e = threadIdx.x; e < n; e += FFT_THREADS) {
        const int r = e / CS, lc = e - r * CS;
        f.store(b, coil, r, c0 + lc, make_float2(0.f, 0.f));
      }
      return;
    }
  }
  for (int e = threadIdx.x; e < n; e += FFT_THREADS) {
    const int r = e / CS, lc = e - r * CS;
    L.buf[e] = f.load(b, coil, r, c0 + lc);
  }
  __syncthreads();
  fft_lines(L, H, CS, 1, CS, true, TWO_WAY ? false : inverse != 0);
  if constexpr (TWO_WAY) {
    for (int e = threadIdx.x; e < n; e += FFT_THREADS) {
      const int r = e / CS, lc = e - r * CS;
      L.buf[e] = f.mid(b, coil, r, c0 + lc, L.buf[e]);
    }
    __syncthreads();
    fft_lines(L, H, CS, 1, CS, true, true);
  }
  for (int e = threadIdx.x; e < n; e += FFT_THREADS) {
    const int r = e / CS, lc = e - r * CS;
    f.store(b, coil, r, c0 + lc, L.buf[e]);
  }
}

// ---- functors ---------------------------------------------------------------------------------------------------
struct ImgGeo {
  int B, H, W;
  __device__ __forceinline__ size_t at(int b, int r, int c) const { return ((size_t)b * H + r) * W + c; }
  __device__ __forceinline__ size_t at(int coil, int b, int r, int c) const {
    return (((size_t)coil * B + b) * H + r) * W + c;
  }
};

// rows of sign * S_c * x -> tmp[coil][b]      (x complex64 or planar; sens NULL: S = 1)
struct RowsFwd {
  ImgGeo g;
  const float2* x;                 // complex input, or NULL ->
  const float *xr, *xi;            // planar input
  const float* sens;
  float2* dst;                     // [coil][b][H][W]
  __device__ __forceinline__ float2 load(int b, int coil, int r, int c) const {
    float2 v;
    if (x) v = x[g.at(b, r, c)];
    else v = make_float2(xr[g.at(b, r, c)], xi[g.at(b, r, c)]);
    float s = sign_rc(r, c);
    if (sens) s *= sens[((size_t)coil * g.H + r) * g.W + c];
    return make_float2(v.x * s, v.y * s);
  }
  __device__ __forceinline__ void store(int b, int coil, int r, int c, float2 v) const { dst[g.at(coil, b, r, c)] = v; }
};

// in-place columns on y[coil][b]: y = mask ? sign * scale * v : 0     (second half of SENSE.__call__ / i2k_complex)
struct ColsFwdMask {
  ImgGeo g;
  float2* y;
  const uint8_t* mask;             // NULL: no mask (plain centred transform)
  int mask_t;
  float scale;
  __device__ __forceinline__ float2 load(int b, int coil, int r, int c) const { return y[g.at(coil, b, r, c)]; }
  __device__ __forceinline__ void store(int b, int coil, int r, int c, float2 v) const {
    const float s = (!mask || mask_at(mask, mask_t, b, g.W, c)) ? sign_rc(r, c) * scale : 0.f;
    y[g.at(coil, b, r, c)] = make_float2(v.x * s, v.y * s);
  }
  __device__ __forceinline__ bool skip(int, int, int) const { return false; }
  __device__ __forceinline__ float2 mid(int, int, int, int, float2 v) const { return v; }
};

// columns of sign * (mask?) s[coil][b] -> tmp[coil][b]   (first half of the adjoint; inverse transform)
struct ColsInvFromS {
  ImgGeo g;
  const float2* s;
  float2* dst;
  const uint8_t* mask;
  int mask_t, apply_mask;
  __device__ __forceinline__ float2 load(int b, int coil, int r, int c) const {
    float sg = sign_rc(r, c);
    if (apply_mask && !mask_at(mask, mask_t, b, g.W, c)) sg = 0.f;
    const float2 v = s[g.at(coil, b, r, c)];
    return make_float2(v.x * sg, v.y * sg);
  }
  __device__ __forceinline__ void store(int b, int coil, int r, int c, float2 v) const { dst[g.at(coil, b, r, c)] = v; }
  __device__ __forceinline__ bool skip(int, int, int) const { return false; }
  __device__ __forceinline__ float2 mid(int, int, int, int, float2 v) const { return v; }
};

// in-place columns on tmp[coil][b]: forward, data-consistency operator in k-space, inverse.
//   mode < 0  SENSE / single-coil L2Penalty residual:  m ? (scale*v - sign*y) : 0
//   mode 1    SingleCoil closed form:                  (scale*v + coef*sign*y) / (1 + coef*m)
//   mode 2    projection:                              coef*sign*y + (m ? 1 - coef : 1) * scale*v
struct ColsProx {
  ImgGeo g;
  float2* tmp;
  const float2* y;                 // [coil][b][H][W]
  const uint8_t* mask;
  int mask_t, mode;
  float scale, coef_host;
  const ipdm_sched_t* sched;       // device schedule: overrides coef_host (modes 1, 2 use it in k-space)
  __device__ __forceinline__ float2 load(int b, int coil, int r, int c) const { return tmp[g.at(coil, b, r, c)]; }
  __device__ __forceinline__ void store(int b, int coil, int r, int c, float2 v) const { tmp[g.at(coil, b, r, c)] = v; }
  __device__ __forceinline__ bool skip(int b, int c0, int cs) const {
    if (mode > 0) return false;
    for (int c = c0; c < c0 + cs; ++c)
      if (mask_at(mask, mask_t, b, g.W, c)) return false;
    return true;
  }
  __device__ __forceinline__ float2 mid(int b, int coil, int r, int c, float2 v) const {
    v.x *= scale;
    v.y *= scale;
    const bool m = mask_at(mask, mask_t, b, g.W, c);
    const float2 yy = y[g.at(coil, b, r, c)];
    const float sg = sign_rc(r, c);
    if (mode <= 0) return m ? make_float2(v.x - sg * yy.x, v.y - sg * yy.y) : make_float2(0.f, 0.f);
    const float coef = sched ? sched->coef : coef_host;
    if (mode == 1) {
      const float inv = m ? 1.f / (1.f + coef) : 1.f;
      return make_float2((v.x + coef * sg * yy.x) * inv, (v.y + coef * sg * yy.y) * inv);
    }
    const float keep = m ? 1.f - coef : 1.f;
    return make_float2(coef * sg * yy.x + keep * v.x, coef * sg * yy.y + keep * v.y);
  }
};

// ---- rows inverse + coil sum + final operation ------------------------------------------------------------------------
enum { FIN_ADJOINT = 0, FIN_SSOS = 1, FIN_L2 = 2, FIN_REPLACE = 3 };

// tmp[coil][b] rows -> inverse row FFT -> acc += sign*scale*S_c * v  (coil order) ->
//   FIN_ADJOINT: out_c[b] = acc        FIN_SSOS: out_f[b] = sqrt(sum |scale*v|^2)
//   FIN_L2: x = x - coef*acc (planar, in place)      FIN_REPLACE: x = acc (planar)
template <int FIN>
__global__ __launch_bounds__(FFT_THREADS) void rows_inv_accum_kernel(const float2* __restrict__ tmp,
                                                                     const float* __restrict__ sens, float2* out_c,
                                                                     float* out_f, float* x_re, float* x_im,
                                                                     const ipdm_sched_t* __restrict__ sched, float coef,
                                                                     int B, int n_coils, int H, int W) {
  STRIP_LDS_SETUP(W)
  if (sched) coef = sched->coef;
  const ImgGeo g{B, H, W};
  const int RS = min(H, STRIP_ELEMS / W);
  const int r0 = blockIdx.x * RS, b = blockIdx.y;
  const int n = RS * W;
  const float scale = rsqrtf((float)H * (float)W);
  float2 acc[EPT];
#pragma unroll
  for (int k = 0; k < EPT; ++k) acc[k] = make_float2(0.f, 0.f);
  for (int coil = 0; coil < n_coils; ++coil) {
#pragma unroll
    for (int k = 0; k < EPT; ++k) {
      const int e = threadIdx.x + k * FFT_THREADS;
      if (e < n) {
        const int lr = e / W, c = e - lr * W;
        L.buf[e] = tmp[g.at(coil, b, r0 + lr, c)];
      }
    }
    __syncthreads();
    fft_lines(L, W, 1, W, RS, false, true);
#pragma unroll
    for (int k = 0; k < EPT; ++k) {
      const int e = threadIdx.x + k * FFT_THREADS;
      if (e < n) {
        const int lr = e / W, c = e - lr * W;
        const float2 v = L.buf[e];
        if constexpr (FIN == FIN_SSOS) {
          acc[k].x += (v.x * v.x + v.y * v.y) * (scale * scale);
        } else {
          float w = sign_rc(r0 + lr, c) * scale;
          if (sens) w *= sens[((size_t)coil * H + r0 + lr) * W + c];
          acc[k].x += v.x * w;
          acc[k].y += v.y * w;
        }
      }
    }
    __syncthreads();
  }
#pragma unroll
  for (int k = 0; k < EPT; ++k) {
    const int e = threadIdx.x + k * FFT_THREADS;
    if (e < n) {
      const int lr = e / W, c = e - lr * W;
      const size_t gi = g.at(b, r0 + lr, c);
      if constexpr (FIN == FIN_ADJOINT) out_c[gi] = acc[k];
      else if constexpr (FIN == FIN_SSOS) out_f[gi] = sqrtf(acc[k].x);
      else if constexpr (FIN == FIN_L2) {
        x_re[gi] = x_re[gi] - coef * acc[k].x;
        x_im[gi] = x_im[gi] - coef * acc[k].y;
      } else {
        x_re[gi] = acc[k].x;
        x_im[gi] = acc[k].y;
      }
    }
  }
}

// Langevin update of both planes in place: x += step*g + noise_scale*n (injected noise or Philox keyed exactly as the
// fused 128x128 kernel keys it: (seed, global sample id, step, plane 0 = real / 1 = imaginary, quad))
__global__ __launch_bounds__(256) void langevin_planes_kernel(float* x_re, float* x_im, const float* __restrict__ g_re,
                                                              const float* __restrict__ g_im, const float* __restrict__ n_re,
                                                              const float* __restrict__ n_im, float step, float noise_scale,
                                                              uint64_t seed, int64_t sample_offset, int64_t step_id,
                                                              const ipdm_sched_t* __restrict__ sched, int HW) {
  if (sched) {
    step = sched->step;
    noise_scale = sched->noise_scale;
    step_id = sched->step_id;
  }
  const int b = blockIdx.y;
  const int quads = HW / 4;                                    // HW is a multiple of 4 (power-of-two images)
  for (int q = blockIdx.x * 256 + threadIdx.x; q < quads; q += gridDim.x * 256) {
    const size_t gi = (size_t)b * HW + 4 * (size_t)q;
    float nr[4], ni[4];
    if (n_re) {
      const float4 a = *reinterpret_cast<const float4*>(n_re + gi), c = *reinterpret_cast<const float4*>(n_im + gi);
      nr[0] = a.x; nr[1] = a.y; nr[2] = a.z; nr[3] = a.w;
      ni[0] = c.x; ni[1] = c.y; ni[2] = c.z; ni[3] = c.w;
    } else {
      ipdm_philox_normal4(seed, sample_offset + b, step_id, 0, (uint32_t)q, nr);
      ipdm_philox_normal4(seed, sample_offset + b, step_id, 1, (uint32_t)q, ni);
    }
    float4 xr = *reinterpret_cast<float4*>(x_re + gi), xi = *reinterpret_cast<float4*>(x_im + gi);
    const float4 gr = *reinterpret_cast<const float4*>(g_re + gi), gim = *reinterpret_cast<const float4*>(g_im + gi);
    xr.x = xr.x + step * gr.x + nr[0] * noise_scale; xr.y = xr.y + step * gr.y + nr[1] * noise_scale;
    xr.z = xr.z + step * gr.z + nr[2] * noise_scale; xr.w = xr.w + step * gr.w + nr[3] * noise_scale;
    xi.x = xi.x + step * gim.x + ni[0] * noise_scale; xi.y = xi.y + step * gim.y + ni[1] * noise_scale;
    xi.z = xi.z + step * gim.z + ni[2] * noise_scale; xi.w = xi.w + step * gim.w + ni[3] * noise_scale;
    *reinterpret_cast<float4*>(x_re + gi) = xr;
    *reinterpret_cast<float4*>(x_im + gi) = xi;
  }
}

// ---- host side --------------------------------------------------------------------------------------------------------
// the dynamic-LDS limit is raised ONCE per kernel instantiation, to the largest strip any call can ask for (a static flag per
// template instance: no host API call on the per-step launch path, none during hipGraph capture after the first use)
template <typename K>
static int set_lds(K kernel, size_t bytes) {
  static bool done = false;                      // one flag per K (per kernel instantiation)
  if (done || bytes <= 64 * 1024) return IPDM_OK;
  hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                                     (int)strip_lds_bytes(2048));        // 2048: the largest side large_ok() admits
  if (e != hipSuccess) return (int)e;
  done = true;
  return IPDM_OK;
}

template <class F>
static int launch_rows(const F& f, int B, int coils, int H, int W, int inverse, hipStream_t s) {
  const size_t lds = strip_lds_bytes(W);
  int rc = set_lds(rows_kernel<F>, lds);
  if (rc) return rc;
  const int RS = H < STRIP_ELEMS / W ? H : STRIP_ELEMS / W;
  hipLaunchKernelGGL(rows_kernel<F>, dim3(H / RS, B, coils), dim3(FFT_THREADS), lds, s, f, H, W, inverse);
  return ipdm_launch_status();
}

template <class F, bool TWO_WAY>
static int launch_cols(const F& f, int B, int coils, int H, int W, int inverse, hipStream_t s) {
  const size_t lds = strip_lds_bytes(H);
  int rc = set_lds(cols_kernel<F, TWO_WAY>, lds);
  if (rc) return rc;
  const int CS = W < STRIP_ELEMS / H ? W : STRIP_ELEMS / H;
  hipLaunchKernelGGL((cols_kernel<F, TWO_WAY>), dim3(W / CS, B, coils), dim3(FFT_THREADS), lds, s, f, H, W, inverse);
  return ipdm_launch_status();
}

template <int FIN>
static int launch_accum(const float2* tmp, const float* sens, float2* out_c, float* out_f, float* x_re, float* x_im,
                        const ipdm_sched_t* sched, float coef, int B, int coils, int H, int W, hipStream_t s) {
  const size_t lds = strip_lds_bytes(W);
  int rc = set_lds(rows_inv_accum_kernel<FIN>, lds);
  if (rc) return rc;
  const int RS = H < STRIP_ELEMS / W ? H : STRIP_ELEMS / W;
  hipLaunchKernelGGL(rows_inv_accum_kernel<FIN>, dim3(H / RS, B), dim3(FFT_THREADS), lds, s, tmp, sens, out_c, out_f, x_re,
                     x_im, sched, coef, B, coils, H, W);
  return ipdm_launch_status();
}

int64_t workspace_bytes(int B, int n_coils, int H, int W) {
  return large_ok(H, W) ? (int64_t)n_coils * B * H * W * (int64_t)sizeof(float2) : 0;
}

// centred orthonormal 2-D (i)FFT, out may alias in: rows into `out`, columns in place on `out`
int fft2c(const float2* in, float2* out, int batch, int H, int W, int inverse, hipStream_t s) {
  const ImgGeo g{batch, H, W};
  RowsFwd rf{g, in, nullptr, nullptr, nullptr, out};
  int rc = launch_rows(rf, batch, 1, H, W, inverse, s);
  if (rc) return rc;
  ColsFwdMask cf{g, out, nullptr, 1, 1.f / sqrtf((float)H * (float)W)};
  return launch_cols<ColsFwdMask, false>(cf, batch, 1, H, W, inverse, s);
}

int sense_forward(const float2* x, const float* sens, const uint8_t* mask, int mask_t, float2* y, int B, int n_coils,
                  int H, int W, hipStream_t s) {
  const ImgGeo g{B, H, W};
  RowsFwd rf{g, x, nullptr, nullptr, sens, y};
  int rc = launch_rows(rf, B, n_coils, H, W, 0, s);
  if (rc) return rc;
  ColsFwdMask cf{g, y, mask, mask_t, 1.f / sqrtf((float)H * (float)W)};
  return launch_cols<ColsFwdMask, false>(cf, B, n_coils, H, W, 0, s);
}

int sense_adjoint(const float2* sm, const float* sens, const uint8_t* mask, int mask_t, int apply_mask, float2* x_out,
                  float* ssos_out, float2* ws, int B, int n_coils, int H, int W, hipStream_t s) {
  const ImgGeo g{B, H, W};
  ColsInvFromS ci{g, sm, ws, mask, mask_t, apply_mask};
  int rc = launch_cols<ColsInvFromS, false>(ci, B, n_coils, H, W, 1, s);
  if (rc) return rc;
  if (ssos_out)
    return launch_accum<FIN_SSOS>(ws, nullptr, nullptr, ssos_out, nullptr, nullptr, nullptr, 0.f, B, n_coils, H, W, s);
  return launch_accum<FIN_ADJOINT>(ws, sens, x_out, nullptr, nullptr, nullptr, nullptr, 0.f, B, n_coils, H, W, s);
}

// Langevin (optional) + data-consistency operator on planar x (in place).  sens NULL = single coil; mode as ColsProx.
int prox_step(float* x_re, float* x_im, const float* g_re, const float* g_im, const float* n_re, const float* n_im,
              float step, float noise_scale, uint64_t seed, int64_t sample_offset, int64_t step_id,
              const ipdm_sched_t* sched, const float2* y, const float* sens, const uint8_t* mask, int mask_t, float coef,
              int mode, float2* ws, int B, int n_coils, int H, int W, hipStream_t s) {
  const int HW = H * W;
  if (g_re) {
    int gx = (HW / 4 + 255) / 256;
    if (gx > 256) gx = 256;
    hipLaunchKernelGGL(langevin_planes_kernel, dim3(gx, B), dim3(256), 0, s, x_re, x_im, g_re, g_im, n_re, n_im, step,
                       noise_scale, seed, (long long)sample_offset, (long long)step_id, sched, HW);
    int rc = ipdm_launch_status();
    if (rc) return rc;
  }
  // (the 128x128 kernel returns early when coef == 0; here the schedule value lives on the device, so the chain always
  //  runs -- with coef == 0 it adds exactly zero for the L2 modes)
  const ImgGeo g{B, H, W};
  RowsFwd rf{g, nullptr, x_re, x_im, sens, ws};
  int rc = launch_rows(rf, B, n_coils, H, W, 0, s);
  if (rc) return rc;
  ColsProx cp{g, ws, y, mask, mask_t, mode, 1.f / sqrtf((float)H * (float)W), coef, sched};
  rc = launch_cols<ColsProx, true>(cp, B, n_coils, H, W, 0, s);
  if (rc) return rc;
  if (mode <= 0) return launch_accum<FIN_L2>(ws, sens, nullptr, nullptr, x_re, x_im, sched, coef, B, n_coils, H, W, s);
  return launch_accum<FIN_REPLACE>(ws, sens, nullptr, nullptr, x_re, x_im, nullptr, 0.f, B, n_coils, H, W, s);
}

}  // namespace ipdm_kspace_large
